#!/bin/bash
out=gpurun_out/r03e
mkdir -p $out
( timeout 1500 python -m pytest tests/test_gpu_field_ntt.py tests/test_gpu_prove.py tests/test_gpu_synth.py tests/test_gpu_multi.py tests/test_gpu_scale.py tests/test_gpu_zkgen.py -m gpu -x -q 2>&1 | tail -30 ) > $out/pytest.txt
cat $out/pytest.txt
for which in tree old; do
  if [ $which = tree ]; then unset ZKHIP_LIB; else export ZKHIP_LIB=$PWD/tools/_ab/libzkhip_$which.so; fi
  ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial $which: ', d['ms_per_step'], d['stage_ms'])"
  python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'], 'ntt', d['stage_ms']['ntt_chain_wall'])"
done > $out/ab.txt 2>&1
cat $out/ab.txt
