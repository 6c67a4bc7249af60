#!/bin/bash
# lanes (4 / 8) at 2^14 ... 2^19 with the wave priorities in place (the eight-lane rule was set before them)
export TMPDIR=/tmp
o=gpurun_out/r04bm; mkdir -p $o
( for rep in 1 2; do for k in 14 16 17 18 19; do for ln in 4 8; do
    ZKHIP_LANES=$ln python bench.py --log2n $k --steps 300 --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k, $ln lanes: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'])"
  done; done; done ) > $o/lanes_with_priorities.txt 2>&1
cat $o/lanes_with_priorities.txt
