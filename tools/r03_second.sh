#!/bin/bash
# round-3 second GPU pass: three builds of the arithmetic (per-MAD asm = tree, C column sums, asm blocks)
mkdir -p gpurun_out/r03b
for p in mul_rate_probe_blk mul_rate_probe_c; do timeout 120 tools/$p; done > gpurun_out/r03b/mul_rate.txt 2>&1
for v in c blk; do
  echo "== tests with libzkhip_$v.so"
  ZKHIP_LIB=$PWD/tools/_ab/libzkhip_$v.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_field_ntt.py -m gpu -x -q 2>&1 | tail -3
done > gpurun_out/r03b/pytest.txt 2>&1
for rep in 1 2; do
for which in tree c blk old; do
  if [ $which = tree ]; then unset ZKHIP_LIB; else export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_$which.so; fi
  python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'g1', d['stage_ms']['g1_l1_kernel'], 'g2', d['stage_ms']['g2_l1_kernel'])"
done
done > gpurun_out/r03b/ab.txt 2>&1
cat gpurun_out/r03b/mul_rate.txt gpurun_out/r03b/pytest.txt gpurun_out/r03b/ab.txt
