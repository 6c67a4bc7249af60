#!/bin/bash
# the same for the middle sizes (2^20, 2^21: A, B1, C as separate launches): B2's merges + reduction on the finishing stream
export TMPDIR=/tmp
o=gpurun_out/r04ar; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2 3; do for k in 19 20 21; do for mx in 0 30; do
    ZKHIP_G2_ASIDE_MAXLOG=$mx python tools/lone_proof.py $k 12 2>/dev/null | awk '/lone proof/ {print $4}' | sort -n | awk -v mx=$mx -v k=$k '{v[NR]=$1} END {printf "2^%d, B2 follow-ups on the finishing stream %s: median %.3f ms, fastest %.3f, slowest %.3f (12 synchronous proofs)\n", k, mx ? "yes" : "no ", (v[6]+v[7])/2, v[1], v[NR]}'
  done; done; done ) > $o/g2_aside_middle_sizes.txt 2>&1
cat $o/g2_aside_middle_sizes.txt
