#!/bin/bash
# default library with: B2 follow-ups of a lone proof on the finishing stream (<= 2^19), the (r, s) part of the tail before the wait
export TMPDIR=/tmp
o=gpurun_out/r04ao; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -3 $o/pytest_gpu.log
( for rep in 1 2; do for k in 14 16 18 20 22; do
    python tools/lone_proof.py $k 16 2>/dev/null | awk '/lone proof/ {print $4}' | sort -n | awk -v k=$k '{v[NR]=$1} END {printf "2^%d: median %.3f ms, fastest %.3f, slowest %.3f (16 synchronous proofs, host witness)\n", k, (v[8]+v[9])/2, v[1], v[NR]}'
  done; done ) > $o/lone_medians.txt 2>&1
cat $o/lone_medians.txt
