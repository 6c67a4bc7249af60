#!/bin/bash
# A/B of the merge levels (wave-parallel segmented scan vs one lane per 32 slots) on one box
for mode in wave serial; do
  if [ $mode = serial ]; then export ZKHIP_LN_SERIAL=1; else unset ZKHIP_LN_SERIAL; fi
  echo "== merge levels: $mode"
  python bench.py --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'], 'join_wait', d['stage_ms']['join_wait'])"
  python bench.py --steps 12 --warmup 3 --no-cpu --witness realistic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' realistic witness: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'])"
  python bench.py --steps 12 --warmup 3 --no-cpu --log2n 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' 2^20: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'])"
  python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world
done
