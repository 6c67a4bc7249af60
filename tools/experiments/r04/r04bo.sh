#!/bin/bash
# sixteen witnesses per submission (ZK_MAX_BATCH 8 -> 16) against eight at the small sizes
export TMPDIR=/tmp
o=gpurun_out/r04bo; mkdir -p $o
( for rep in 1 2; do for k in 14 15 16; do for b in 8 16; do
    python bench.py --log2n $k --batch $b --steps 480 --warmup 16 --no-cpu 2>&1 | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k, $b per submission:', d['ms_per_step'], 'ms per proof,', round(d['value'],1), 'proofs/s; in flight', d['config']['proofs_in_flight'], '; window', d['config']['window_bits'])" 2>&1 | tail -1
  done; done; done ) > $o/batch16.txt 2>&1
cat $o/batch16.txt
