#!/bin/bash
# round 5, the final code: full GPU suite, then every number DESIGN.md quotes from ONE box (tools/round_numbers.sh)
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05z_tests.txt
bash tools/round_numbers.sh r05z > gpurun_out/r05z_round_numbers.txt 2>&1
cp gpurun_out/r05z_tests.txt gpurun_out/r05z/profiles/r05z_gpu_tests.txt
cat gpurun_out/r05z_tests.txt; tail -30 gpurun_out/r05z_round_numbers.txt
