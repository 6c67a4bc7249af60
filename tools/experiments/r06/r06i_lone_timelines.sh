mkdir -p gpurun_out/r06i
export TMPDIR=/tmp
R=$PWD
cd /tmp
for k in 20 21 22; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r06i/t$k -o t -- python $R/tools/lone_proof.py $k 4 > $R/gpurun_out/r06i/lone$k.log 2>&1
  python $R/tools/lone_timeline.py $R/gpurun_out/r06i/t$k 80 > $R/gpurun_out/r06i/timeline_2p$k.txt 2>&1
  grep "^lone" $R/gpurun_out/r06i/lone$k.log >> $R/gpurun_out/r06i/timeline_2p$k.txt
  rm -rf $R/gpurun_out/r06i/t$k
done
cat $R/gpurun_out/r06i/timeline_2p20.txt $R/gpurun_out/r06i/timeline_2p21.txt
