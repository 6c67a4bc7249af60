#!/bin/bash
# round 5, the final code: full GPU suite, then every number DESIGN.md quotes from ONE box (tools/round_numbers.sh)
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05m_tests.txt
bash tools/round_numbers.sh r05m > gpurun_out/r05m_round_numbers.txt 2>&1
cp gpurun_out/r05m_tests.txt gpurun_out/r05m/profiles/r05m_gpu_tests.txt
cat gpurun_out/r05m_tests.txt; tail -30 gpurun_out/r05m_round_numbers.txt
