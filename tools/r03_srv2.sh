#!/bin/bash
mkdir -p gpurun_out/r03o
for k in 14 16; do
  for route in input witness; do timeout 300 python tools/server_bench.py $k 1024 0 $route 2>&1 | tail -1; done
  timeout 300 python bench.py --log2n $k --batch 4 --steps 512 --warmup 16 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C-ABI loop 2^$k batch 4:', d['value'], 'proofs/s')"
done > gpurun_out/r03o/server.txt 2>&1
cat gpurun_out/r03o/server.txt
