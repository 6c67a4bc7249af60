#!/bin/bash
out=gpurun_out/r03h
mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_field_ntt.py tests/test_gpu_prove.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -4 ) > $out/pytest.txt
cat $out/pytest.txt
bash tools/ntt_counters.sh r03h_cnt 22 > /dev/null 2>&1
grep "calls" gpurun_out/r03h_cnt/summary.txt
export TMPDIR=/tmp
echo "== split in two outer groups (probes build)"
ZKHIP_LIB=$PWD/rapidsnark-old_amd/libzkhip_probes.so ZKHIP_NTT_SPLIT=2 rocprofv3 --kernel-trace --stats --output-format csv -d $out/split -o s -- python tools/ntt_probe.py 22 3 > $out/split.log 2>&1
python - $out/split <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ntt" in r["Name"] or "abc_to_h" in r["Name"]:
            print("%-70s calls %s avg_ns %s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
PY
for k in 20 24; do
echo "== 2^$k"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k$k -o s -- python tools/ntt_probe.py $k 3 > $out/k$k.log 2>&1
python - $out/k$k <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ntt" in r["Name"] or "abc_to_h" in r["Name"]:
            print("%-70s calls %s avg_ns %s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
PY
done
