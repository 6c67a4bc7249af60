#!/bin/bash
# second look, MEDIAN of 16 lone proofs per run (the means of r04am carry one or two slow proofs each)
export TMPDIR=/tmp
o=gpurun_out/r04an; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2 3; do for k in 14 15 16 17 18 19; do for mx in 0 30; do
    ZKHIP_G2_ASIDE_MAXLOG=$mx python tools/lone_proof.py $k 16 2>/dev/null | awk '/lone proof/ {print $4}' | sort -n | awk -v mx=$mx -v k=$k '{v[NR]=$1} END {printf "2^%d, B2 follow-ups on the finishing stream %s: median %.3f ms, fastest %.3f, slowest %.3f (16 synchronous proofs)\n", k, mx ? "yes" : "no ", (v[8]+v[9])/2, v[1], v[NR]}'
  done; done; done ) > $o/g2_aside_medians.txt 2>&1
cat $o/g2_aside_medians.txt
