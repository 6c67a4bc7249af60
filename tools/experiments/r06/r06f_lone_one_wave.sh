# premise check: do the transform chain and sort(h) of a LONE proof run beside level-1 launches held to ONE workgroup per CU
# (ZKHIP_ACC_ROUND_WGS=1, probes build: one level-1 wave per SIMD, 154 / 212 VGPRs, leaves room for a 252-VGPR transform wave)?
mkdir -p gpurun_out/r06f
export TMPDIR=/tmp
R=$PWD
export ZKHIP_LIB=$R/rapidsnark-old_amd/libzkhip_probes.so
cd /tmp
for wgs in 1 0; do
  if [ $wgs = 0 ]; then unset ZKHIP_ACC_ROUND_WGS; else export ZKHIP_ACC_ROUND_WGS=$wgs; fi
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r06f/t$wgs -o t -- python $R/tools/lone_proof.py 22 4 > $R/gpurun_out/r06f/lone_wgs$wgs.log 2>&1
  python $R/tools/lone_timeline.py $R/gpurun_out/r06f/t$wgs 100 > $R/gpurun_out/r06f/timeline_wgs$wgs.txt 2>&1
  grep "^lone" $R/gpurun_out/r06f/lone_wgs$wgs.log >> $R/gpurun_out/r06f/timeline_wgs$wgs.txt
  rm -rf $R/gpurun_out/r06f/t$wgs
done
cat $R/gpurun_out/r06f/timeline_wgs1.txt
