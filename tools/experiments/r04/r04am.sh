#!/bin/bash
# lone proofs: merges + reduction of MSM B2 on the idle finishing stream (probes build: ZKHIP_G2_ASIDE_MAXLOG = largest log2 size it applies to)
export TMPDIR=/tmp
o=gpurun_out/r04am; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for rep in 1 2 3; do for k in 14 16 18 19 22; do for mx in 0 30; do
    ZKHIP_G2_ASIDE_MAXLOG=$mx python tools/lone_proof.py $k 8 2>/dev/null | awk -v mx=$mx -v k=$k '/lone proof/ {s+=$4; n++} END {printf "2^%d, B2 follow-ups on the finishing stream %s: %.3f ms per synchronous proof (mean of %d)\n", k, mx ? "yes" : "no ", s/n, n}'
  done; done; done ) > $o/g2_aside.txt 2>&1
cat $o/g2_aside.txt
