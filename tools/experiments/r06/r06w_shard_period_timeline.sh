mkdir -p gpurun_out/r06w
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06w/t -o t -- python $R/tools/shard_lone.py 22 8 16 2 > $R/gpurun_out/r06w/run.txt 2>&1
python $R/tools/timeline.py $R/gpurun_out/r06w/t 30 10 > $R/gpurun_out/r06w/period_timeline.txt 2>&1
rm -rf $R/gpurun_out/r06w/t
cat $R/gpurun_out/r06w/period_timeline.txt
