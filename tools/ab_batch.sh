#!/bin/bash
# MSM A, B1, C in one set of launches (ZKHIP_BATCH_ABC) vs separately, same box
for k in 14 16 18 20; do for b in 1 0; do
  ZKHIP_BATCH_ABC=$b python bench.py --log2n $k --steps $((k<20?80:20)) --warmup 8 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$k batch=$b: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
done; done
for b in 1 0; do ZKHIP_BATCH_ABC=$b python tools/server_bench.py 16 256 0 2>/dev/null; done
