#!/bin/bash
# full GPU suite on the new defaults (tail pool, eight lanes up to 2^16, server batch 8 up to 2^16) + the REST throughput table
export TMPDIR=/tmp
o=gpurun_out/r04ak; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -3 $o/pytest_gpu.log
( for k in 14 15 16; do for b in 4 8; do
    ZKHIP_BATCH=$b python tools/server_bench.py $k 2048 0 witness 2>/dev/null | sed "s/^/ZKHIP_BATCH=$b /"
  done; python tools/server_bench.py $k 2048 0 input 2>/dev/null | sed "s/^/default /"; done ) > $o/server_throughput.txt 2>&1
cat $o/server_throughput.txt | cut -c1-260
