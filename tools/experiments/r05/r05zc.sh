#!/bin/bash
# strided first level of the bucket reduction (k_msm_reduce_strided) against the chunked form alone: parity tests, then same-box A/B
export TMPDIR=/tmp
python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_scale.py -q -m gpu -x 2>&1 | tail -4
out=gpurun_out/r05zc_ab.txt; : > $out
echo "2^22" >> $out; REPS=3 bash tools/ab_lib.sh old >> $out
echo "2^20" >> $out; REPS=3 bash tools/ab_lib.sh old --log2n 20 >> $out
echo "2^21" >> $out; REPS=2 bash tools/ab_lib.sh old --log2n 21 >> $out
cat $out
