#!/bin/bash
# the round's final numbers from one box: full GPU suite, then tools/round_numbers.sh (driver's command under the profilers, per-leg
# kernel tables, PMC traffic, instruction budget, timelines, the other sizes, shards, server, CLI)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04at/pytest_gpu.log 2>&1; tail -4 gpurun_out/r04at/pytest_gpu.log
bash tools/round_numbers.sh r04at > gpurun_out/r04at/round_numbers.log 2>&1
tail -40 gpurun_out/r04at/round_numbers.log
cat gpurun_out/r04at/profiles/r04at_legs.txt
