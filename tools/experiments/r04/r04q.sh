#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04q
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04q/pytest_gpu.log 2>&1; tail -4 gpurun_out/r04q/pytest_gpu.log
bash tools/round_numbers.sh r04q > gpurun_out/r04q/round_numbers.log 2>&1
tail -40 gpurun_out/r04q/round_numbers.log
cat gpurun_out/r04q/profiles/r04q_legs.txt
