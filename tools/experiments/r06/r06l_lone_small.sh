mkdir -p gpurun_out/r06l
export TMPDIR=/tmp
R=$PWD
cd /tmp
for k in 16 18; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r06l/t$k -o t -- python $R/tools/lone_proof.py $k 4 > $R/gpurun_out/r06l/lone$k.log 2>&1
  python $R/tools/lone_timeline.py $R/gpurun_out/r06l/t$k 8 > $R/gpurun_out/r06l/timeline_2p$k.txt 2>&1
  grep "^lone" $R/gpurun_out/r06l/lone$k.log >> $R/gpurun_out/r06l/timeline_2p$k.txt
  rm -rf $R/gpurun_out/r06l/t$k
done
