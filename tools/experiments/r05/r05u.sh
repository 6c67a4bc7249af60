#!/bin/bash
# where do the cycles of the sort kernels go?  counters over a SERIAL run (ZKHIP_SERIAL=1: one stream, kernels alone on the chip)
export TMPDIR=/tmp ZKHIP_SERIAL=1
out=gpurun_out/r05u; mkdir -p $out
cmd="python bench.py --steps 2 --warmup 1 --in-flight 1 --pipeline 0 --no-cpu --no-counters"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o c -- $cmd > $out/p$i.log 2>&1 || echo "group '$grp' failed" >> $out/err.txt
done
python - $out <<'PY' > $out/sort_counters.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void zk::", "").replace("zk::", "").split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].replace("void zk::", "").replace("zk::", "").split("(")[0][:48]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    if not any(x in k for x in ("k_bin", "k_scan", "k_msm_digits", "k_msm_compact", "k_spmv", "k_abc_to_h")):
        continue
    print(k, " launches", len(dur[k]), " mean_us %.1f  (under counters, serial)" % (sum(dur[k]) / max(1, len(dur[k])) / 1e3))
    for c in sorted(acc[k]):
        print("   %-24s %.4g per launch" % (c, acc[k][c] / cnt[k][c]))
PY
cat $out/sort_counters.txt
