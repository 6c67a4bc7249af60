#!/bin/bash
# round 5, the final code (split bucket reduction, c = 20 at 2^21, busy-mode lanes, A|B1|C in one launch at 2^21): full GPU suite, then every number DESIGN.md quotes from ONE box (tools/round_numbers.sh)
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05zr_tests.txt
bash tools/round_numbers.sh r05zr > gpurun_out/r05zr_round_numbers.txt 2>&1
cp gpurun_out/r05zr_tests.txt gpurun_out/r05zr/profiles/r05zr_gpu_tests.txt
cat gpurun_out/r05zr_tests.txt; tail -30 gpurun_out/r05zr_round_numbers.txt
(python tools/soak_mixed.py; python tools/soak.py 20 30; python tools/soak.py 21 20; python tools/soak.py 22 30) 2>&1 | grep -v "amdgpu.ids" | grep -i "proofs\|mismatch\|error\|Traceback" > gpurun_out/r05zr/profiles/r05zr_soak.txt
cat gpurun_out/r05zr/profiles/r05zr_soak.txt
