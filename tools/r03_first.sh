#!/bin/bash
# round-3 first GPU pass: correctness of the new arithmetic, product-rate probes, old-vs-new A/B
mkdir -p gpurun_out/r03a
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03a/pytest.txt
for p in mul_rate_probe mul_rate_probe_serial mul_rate_probe_c; do timeout 120 tools/$p; done > gpurun_out/r03a/mul_rate.txt 2>&1
timeout 900 bash tools/ab_lib.sh > gpurun_out/r03a/ab.txt 2>&1
cat gpurun_out/r03a/pytest.txt gpurun_out/r03a/mul_rate.txt gpurun_out/r03a/ab.txt
