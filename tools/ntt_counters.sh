#!/bin/bash
# Per-kernel durations and SQ counters of the NTT pipeline alone.  tools/ntt_counters.sh <tag> [k]
tag=$1; k=${2:-22}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cmd="python tools/ntt_probe.py $k 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o s -- $cmd > $out/st.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/c$i -o c -- $cmd > $out/c$i.log 2>&1
done
python - $out <<'PY' > $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ntt" in r["Name"] or "abc_to_h" in r["Name"] or "pair" in r["Name"]:
            print("%-70s calls %s avg_ns %s total_ns %s" % (r["Name"][:70], r["Calls"], r["AverageNs"], r["TotalDurationNs"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/c*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    if "ntt" in k:
        print(k)
        for c in sorted(acc[k]): print("    %-26s %.4g per launch" % (c, acc[k][c] / cnt[k][c]))
PY
cat $out/summary.txt
