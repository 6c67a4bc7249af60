#!/bin/bash
# follow-up streams: five (one per MSM) vs two, same box, several sizes
for t in 5 2; do   # ZKHIP_TAIL: follow-up streams (2 = default)
  export ZKHIP_TAIL=$t
  for k in 14 16 18 20 22; do
    python bench.py --log2n $k --steps $((k<20?60:15)) --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail streams $t  2^$k: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'])"
  done
done
