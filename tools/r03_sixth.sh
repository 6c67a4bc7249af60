#!/bin/bash
out=gpurun_out/r03g
mkdir -p $out
( timeout 1500 python -m pytest tests/test_gpu_field_ntt.py tests/test_gpu_prove.py tests/test_gpu_synth.py tests/test_gpu_multi.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -8 ) > $out/pytest.txt
cat $out/pytest.txt
bash tools/ntt_counters.sh r03g_cnt 22 > /dev/null 2>&1
grep -A3 "calls\|k_ntt" gpurun_out/r03g_cnt/summary.txt | grep "calls\|SQ_WAIT_ANY\|SQ_LDS_BANK\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\|k_ntt" 
for which in tree; do
  ZKHIP_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial $which: ', d['ms_per_step'], d['stage_ms'])"
  python bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which: host', d['ms_per_step'], 'resident', d['resident_witness']['ms_per_step'], 'latency', d['latency_ms_one_at_a_time'], 'ntt', d['stage_ms']['ntt_chain_wall'])"
done
