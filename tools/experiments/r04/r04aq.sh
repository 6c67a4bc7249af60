#!/bin/bash
# the slow lone proofs at 2^17 ... 2^19 (one in ~16 takes 9-10 ms instead of 2.4-7): is it the host-function staging of the pieces?
export TMPDIR=/tmp
o=gpurun_out/r04aq; mkdir -p $o
export ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so
( for k in 16 17 18; do for st in 0 1; do
    if [ $st = 1 ]; then export ZKHIP_STAGE_SYNC=1; else unset ZKHIP_STAGE_SYNC; fi
    python tools/lone_proof.py $k 96 2>/dev/null | awk '/lone proof/ {print $4}' | sort -n | awk -v st=$st -v k=$k '{v[NR]=$1} END {m=(v[48]+v[49])/2; n=0; for(i=1;i<=NR;i++) if (v[i] > 1.5*m) n++; printf "2^%d, staging %s: median %.3f ms, 90th percentile %.3f, slowest %.3f; %d of %d proofs slower than 1.5 x the median\n", k, st ? "inside the call      " : "by a host function   ", m, v[87], v[NR], n, NR}'
  done; done ) > $o/slow_lone_proofs.txt 2>&1
cat $o/slow_lone_proofs.txt
