#!/bin/bash
# lanes (independent stream pairs + a|b|c|h buffers a slot is bound to): 4 (default) against 6 / 8 at the small and middle sizes
export TMPDIR=/tmp
o=gpurun_out/r04aj; mkdir -p $o
( for rep in 1 2; do for cfg in "14 1" "14 4" "16 1" "16 4" "18 1" "20 1"; do set -- $cfg; for ln in 4 6 8; do
    ZKHIP_LANES=$ln python bench.py --log2n $1 --steps $([ $1 -ge 20 ] && echo 40 || echo 240) --warmup 8 --batch $2 --in-flight 8 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$1, $2 per submission, 8 in flight, lanes $ln:', d['ms_per_step'], 'ms per proof; resident', d['resident_witness']['ms_per_step'])"
  done; done; done ) > $o/lanes.txt 2>&1
cat $o/lanes.txt
