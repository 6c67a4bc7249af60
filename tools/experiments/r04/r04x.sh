#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04x; mkdir -p $o
( python tools/soak.py 20 150 2>&1 | grep -v amdgpu
  python tools/soak.py 16 60 2>&1 | grep -v amdgpu
  python tools/soak.py 22 90 2>&1 | grep -v amdgpu
  python tools/server_bench.py 16 8192 0 witness 2>/dev/null ) > $o/soak.txt 2>&1
cat $o/soak.txt
