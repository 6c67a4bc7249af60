#!/bin/bash
# The N > 1 path of bench.py on a 1-GPU box: every rank on cuda:0, exchange over gloo (ZK_BENCH_SHARE_GPU=1).
# Exercises everything but the RCCL calls themselves; the proof is verified against an unsharded prover.
k=${1:-18}
for n in 2 4 8; do
  for chain in partitioned replicated; do
    echo "== $n ranks, chain $chain, 2^$k"
    ZK_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 4 --warmup 1 --log2n $k --no-cpu --chain $chain 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(' ms/proof', d['ms_per_step'], 'verified', d.get('multi_gpu_proof_equals_single_gpu_proof'), d['config']['parallelism'])"
  done
done
