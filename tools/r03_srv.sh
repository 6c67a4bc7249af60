#!/bin/bash
mkdir -p gpurun_out/r03k
ulimit -c 0
for i in 1 2 3 4 5 6; do
  for b in 1 4; do
    SERVER_LOG=gpurun_out/r03k/srv_${i}_$b.log ZKHIP_BATCH=$b timeout 120 python tools/server_bench.py 15 64 0,0 input 2>&1 | tail -3
    tail -3 gpurun_out/r03k/srv_${i}_$b.log
  done
done
