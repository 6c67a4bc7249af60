"""Condense rocprofv3 outputs under gpurun_out/ into the small files committed under profiles/.

  python tools/summarize_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> <bench_json>

FETCH_SIZE / WRITE_SIZE are collected in SEPARATE --pmc passes (TCC slots: FETCH_SIZE costs 3,
WRITE_SIZE 2 — MI355X_MICROARCH.md §rocprofv3 PMC slots), unit = KiB.  On gfx950 FETCH_SIZE
reports exactly half the bytes of a WIDE COALESCED stream; the MSM gathers are random 64/128-B
reads for which the guide gives no calibration, so both the raw and the x2-corrected figures
are written and the raw one is what bench.py reports (it already exceeds the logical gather volume)."""
import collections
import csv
import json
import os
import shutil
import sys

tag, stats_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def find(d, suffix):
    for f in os.listdir(d):
        if f.endswith(suffix):
            return os.path.join(d, f)
    raise SystemExit("no %s in %s" % (suffix, d))


def per_kernel(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


shutil.copy(find(stats_dir, "kernel_stats.csv"), os.path.join(out, "%s_kernel_stats.csv" % tag))
f = per_kernel(find(fetch_dir, "counter_collection.csv"), "FETCH_SIZE")
w = per_kernel(find(write_dir, "counter_collection.csv"), "WRITE_SIZE")
bench = {}
if os.path.exists(bench_json):
    js = [ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")]
    if js:
        bench = json.loads(js[-1])
# only what identifies the configuration travels with the counters (the bench line's own roofline.traffic is a REPLAY of an earlier
# file of this kind: embedding it here made a file cite its predecessor as its source)
bench = {"config": bench.get("config", {}), "n_gpus": bench.get("n_gpus"), "ms_per_step": bench.get("ms_per_step")}
summary = {"unit": "KiB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes)", "bench": bench, "kernels": {}}
for k in sorted(f, key=lambda k: -sum(f[k])):
    fa = sum(f[k]) / len(f[k])
    wa = sum(w[k]) / len(w[k]) if k in w else 0.0
    summary["kernels"][k] = {"launches_profiled": len(f[k]), "fetch_kib": round(fa, 1), "write_kib": round(wa, 1),
                             "hbm_bytes_raw": int((fa + wa) * 1024), "hbm_bytes_read_x2": int((2 * fa + wa) * 1024)}
json.dump(summary, open(os.path.join(out, "%s_pmc_traffic.json" % tag), "w"), indent=1)
print("wrote profiles/%s_kernel_stats.csv and profiles/%s_pmc_traffic.json" % (tag, tag))
