#!/bin/bash
# round 5, the final code: full GPU suite, then every number DESIGN.md quotes from ONE box (tools/round_numbers.sh)
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05p_tests.txt
bash tools/round_numbers.sh r05p > gpurun_out/r05p_round_numbers.txt 2>&1
cp gpurun_out/r05p_tests.txt gpurun_out/r05p/profiles/r05p_gpu_tests.txt
cat gpurun_out/r05p_tests.txt; tail -30 gpurun_out/r05p_round_numbers.txt
