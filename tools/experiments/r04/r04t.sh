#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04t; mkdir -p $o
python -m pytest tests/test_gpu_synth.py -m gpu -x -q -k "sparse_witness or circuit_shaped or realistic" > $o/pytest_sparse.log 2>&1; tail -3 $o/pytest_sparse.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'one at a time', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'sync', d['ms_per_proof_sync'])"; }
( for rep in 1 2 3; do for sp in 0 1; do
    python bench.py --steps 30 --warmup 3 --no-cpu --witness realistic --shape circuit --sparse-witness $sp 2>/dev/null | line "2^22 circuit-shaped, realistic witness, sparse-witness flag $sp"
  done; done
  for sp in 0 1; do python bench.py --steps 30 --warmup 3 --no-cpu --witness realistic --sparse-witness $sp 2>/dev/null | line "2^22 dense tables, realistic witness, flag $sp"; done
  for sp in 0 1; do python bench.py --steps 30 --warmup 3 --no-cpu --log2n 20 --witness realistic --shape circuit --sparse-witness $sp 2>/dev/null | line "2^20 circuit-shaped realistic, flag $sp"; done
  for sp in 0 1; do python bench.py --steps 16 --warmup 3 --no-cpu --sparse-witness $sp 2>/dev/null | line "2^22 UNIFORM witness (the wrong use of the flag), flag $sp"; done
) > $o/ab_sparse_witness.txt 2>&1
cat $o/ab_sparse_witness.txt
