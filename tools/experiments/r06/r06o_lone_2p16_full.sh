mkdir -p gpurun_out/r06o
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/r06o/t16 -o t -- python $R/tools/lone_proof.py 16 4 > $R/gpurun_out/r06o/lone16.log 2>&1
python $R/tools/lone_timeline.py $R/gpurun_out/r06o/t16 0 > $R/gpurun_out/r06o/timeline_2p16_all.txt 2>&1
ls -la $R/gpurun_out/r06o/t16/*/ | head
# keep the hip api trace of the last proof only (tail)
f=$(find $R/gpurun_out/r06o/t16 -name "*hip_api_trace.csv" | head -1); [ -n "$f" ] && tail -n 400 "$f" > $R/gpurun_out/r06o/hip_api_tail.csv
k=$(find $R/gpurun_out/r06o/t16 -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && tail -n 140 "$k" > $R/gpurun_out/r06o/kernel_tail.csv
rm -rf $R/gpurun_out/r06o/t16
