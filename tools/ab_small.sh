#!/bin/bash
# small circuits: chunk parameters of the level-1 accumulation and of the bucket reduction
run() { env $2 python bench.py --log2n $1 --steps 80 --warmup 10 --no-cpu 2>>gpurun_out/ab_small.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$1 $2: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'])"; }
for k in 14 16 18; do
  run $k "ZKHIP_REDUCE_CHUNK=16 ZKHIP_ACC_CHUNK_MIN=32"
  run $k "ZKHIP_REDUCE_CHUNK=4 ZKHIP_ACC_CHUNK_MIN=32"
  run $k "ZKHIP_REDUCE_CHUNK=4 ZKHIP_ACC_CHUNK_MIN=16"
  run $k "ZKHIP_REDUCE_CHUNK=4 ZKHIP_ACC_CHUNK_MIN=8"
  run $k "ZKHIP_REDUCE_CHUNK=2 ZKHIP_ACC_CHUNK_MIN=8"
done
