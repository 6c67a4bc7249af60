#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04c; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; tail -5 $o/pytest_gpu.log
tools/ab_lone.sh > $o/ab_lone.txt 2>&1; cat $o/ab_lone.txt
tools/ab_sync.sh > $o/ab_sync.txt 2>&1; grep "2^" $o/ab_sync.txt
k=22
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/lone$k -o t -- python tools/lone_proof.py $k 4 > $o/lone$k.log 2>&1
python tools/lone_timeline.py $o/lone$k 100 > $o/lone${k}_timeline.txt 2>&1
grep "^lone" $o/lone$k.log >> $o/lone${k}_timeline.txt
find $o/lone$k -name '*.csv' -size +20M -delete
tail -30 $o/lone22_timeline.txt
