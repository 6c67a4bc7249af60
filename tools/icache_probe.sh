#!/bin/bash
# Instruction-cache behaviour of the level-1 kernels (their loops are 18 KB (G1) and 58 KB (G2) of code; the
# I-cache is 64 KB per two CUs).  tools/icache_probe.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cmd="python bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS" "SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_VALU SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/i$i -o c -- $cmd > $out/i$i.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/i*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    if "accum_l1" in k or "ntt_pass" in k or "reduce_chunks" in k:
        print(k, {c: "%.4g" % (acc[k][c] / cnt[k][c]) for c in sorted(acc[k])})
PY
