#!/bin/bash
# A/B of the NTT workgroup shape on one box: unsharded throughput/latency and the 8-way shard share
for cfg in "512 11" "256 11" "256 10"; do
  set -- $cfg
  export ZKHIP_NTT_THREADS=$1 ZKHIP_NTT_TILE=$2
  echo "== threads $1 tile 2^$2"
  python bench.py --steps 12 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'], 'ntt', d['stage_ms']['ntt_chain_wall'])"
  python tools/shard_probe.py 22 8 partitioned 2>&1 | grep world
  ZKHIP_SERIAL=1 python tools/shard_probe.py 22 1 replicated 2>&1 | grep world
done
