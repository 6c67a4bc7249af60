#!/bin/bash
# A|B1|C as one set of launches at the mid sizes (2^20, 2^21: off by the round's rule) with the wave priorities in place
export TMPDIR=/tmp
o=gpurun_out/r04bn; mkdir -p $o
( for rep in 1 2 3; do for k in 20 21; do for b in 0 1; do
    ZKHIP_BATCH_ABC=$b python bench.py --log2n $k --steps 60 --warmup 4 --no-cpu 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('2^$k, A|B1|C batched $b: period', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'sync', d['ms_per_proof_sync'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], d['config'].get('msm_a_b1_c_in_one_launch'))"
  done; done; done ) > $o/batch_abc_mid_sizes.txt 2>&1
cat $o/batch_abc_mid_sizes.txt
