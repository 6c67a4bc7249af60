#!/bin/bash
mkdir -p gpurun_out/r03c
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r03c/pytest.txt
timeout 120 tools/mul_rate_probe > gpurun_out/r03c/mul_rate.txt 2>&1
for rep in 1 2; do
for which in tree blk old; do
  if [ $which = tree ]; then unset ZKHIP_LIB; else export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_$which.so; fi
  python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time']['witness_in_hbm'], 'g1', d['stage_ms']['g1_l1_kernel'], 'g2', d['stage_ms']['g2_l1_kernel'])"
done
done > gpurun_out/r03c/ab.txt 2>&1
unset ZKHIP_LIB
ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial: ', d['ms_per_step'], d['stage_ms'])" >> gpurun_out/r03c/ab.txt
ZKHIP_LIB=$PWD/tools/_ab/libzkhip_old.so ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial old: ', d['ms_per_step'], d['stage_ms'])" >> gpurun_out/r03c/ab.txt
cat gpurun_out/r03c/pytest.txt gpurun_out/r03c/mul_rate.txt gpurun_out/r03c/ab.txt
