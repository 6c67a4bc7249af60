#!/bin/bash
# Where do the cycles of the G1/G2 L1 accumulation kernels go?  (GPU box, repo root)
#   tools/stall_probe.sh <tag>
# Separate --pmc passes over a serial 2^22 proof run; per-kernel sums written to gpurun_out/<tag>/stall.txt.
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cmd="python bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_IFETCH SQ_INSTS_SMEM" "SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o c -- $cmd > $out/p$i.log 2>&1 || echo "group '$grp' failed" >> $out/stall.err
done
python - $out <<'PY' > $out/stall.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(k, " launches", cnt[k].get("GRBM_GUI_ACTIVE"), " mean_ns", sum(dur[k]) / max(1, len(dur[k])))
    for c in sorted(acc[k]):
        print("   %-28s %.4g per launch" % (c, acc[k][c] / cnt[k][c]))
PY
cat $out/stall.txt | head -120
