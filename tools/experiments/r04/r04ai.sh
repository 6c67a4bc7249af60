#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04ai; mkdir -p $o
( for k in 14 16; do
    python tools/submit_cost.py $k 8
    ZKHIP_GRAPH=1 python tools/submit_cost.py $k 8
    ZKHIP_LANES=8 python tools/submit_cost.py $k 8
    python tools/submit_cost.py $k 2
  done ) > $o/submit_cost.txt 2>&1
cat $o/submit_cost.txt
