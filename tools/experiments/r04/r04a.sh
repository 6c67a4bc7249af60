#!/bin/bash
# round 4, first GPU call: batched-affine probe, lone-proof timelines, this box's baseline
export TMPDIR=/tmp
o=gpurun_out/r04a; mkdir -p $o
tools/batch_affine_probe > $o/batch_affine_probe.txt 2>&1
tools/mul_rate_probe > $o/mul_rate.txt 2>&1
for k in 22 20; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/lone$k -o t -- python tools/lone_proof.py $k 4 > $o/lone$k.log 2>&1
  python tools/lone_timeline.py $o/lone$k 100 > $o/lone${k}_timeline.txt 2>&1
  find $o/lone$k -name '*.csv' -size +20M -delete
done
python bench.py > $o/bench.json 2> $o/bench.err
timeout 600 python -m pytest tests/test_gpu_zkgen.py tests/test_gpu_scale.py -m gpu -x -q > $o/pytest_views.log 2>&1
tail -3 $o/pytest_views.log; cat $o/batch_affine_probe.txt; tail -3 $o/lone22_timeline.txt; cat $o/lone22.log | tail -5
