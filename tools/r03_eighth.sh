#!/bin/bash
out=gpurun_out/r03i
mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_field_ntt.py tests/test_gpu_prove.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -4 ) > $out/pytest.txt
cat $out/pytest.txt
bash tools/ntt_counters.sh r03i_cnt 22 > /dev/null 2>&1
grep "calls" gpurun_out/r03i_cnt/summary.txt
grep -A17 "^zk::k_ntt_mid$" gpurun_out/r03i_cnt/summary.txt | grep "WAIT\|INSTS_VALU\|WAVE_CYC"
