set -x
mkdir -p gpurun_out/r06b
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06b/busy -o busy -- python $R/tools/shard_lone.py 22 8 40 2 > $R/gpurun_out/r06b/busy.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06b/lone -o lone -- python $R/tools/shard_lone.py 22 8 4 1 > $R/gpurun_out/r06b/lone.txt 2>&1
cd $R
python tools/lone_timeline.py gpurun_out/r06b/lone 10 > gpurun_out/r06b/lone_timeline.txt 2>&1
python tools/shard_lone.py 22 8 40 2 > gpurun_out/r06b/busy_plain.txt 2>&1
find gpurun_out/r06b -name "*kernel_stats.csv" | head
find gpurun_out/r06b -name "*.csv" -size +20M -delete
tail -n 3 gpurun_out/r06b/busy.txt; tail -n 3 gpurun_out/r06b/busy_plain.txt
