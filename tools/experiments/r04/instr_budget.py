"""VALU instruction budget of one proof from a rocprofv3 --pmc SQ_INSTS_VALU run (ZKHIP_SERIAL=1 recommended).
    python tools/instr_budget.py <dir> [proofs_profiled: default = the number of k_spmv_abc launches] [bench.json out.json]
With the last two arguments the budget is also written as JSON next to the bench line's configuration — what bench.py
replays as roofline.issue_bound.valu_instructions_per_proof."""
import csv, glob, collections, json, sys
d = sys.argv[1]
nproofs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0      # 0 = count them: one k_spmv_abc launch per proof
agg = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'SQ_INSTS_VALU':
            continue
        n = r['Kernel_Name'].replace('void zk::', '').replace('zk::', '')
        key = n.split('(')[0]
        if 'Fp2T' in n:
            key += ' [G2]'
        agg[key] += float(r['Counter_Value']); cnt[key] += 1
if nproofs <= 0:
    nproofs = float(max(1, cnt.get('k_spmv_abc', 1)))
skip = ('precomp', 'chain', 'build_tables', 'pair_tables', 'fq_to_internal', 'fr_convert', 'csr', 'fixed_base')
rows = [(v / nproofs, k, cnt[k] / nproofs) for k, v in agg.items() if not any(x in k for x in skip)]
tot = sum(v for v, _, _ in rows)
for v, k, c in sorted(rows, reverse=True)[:18]:
    print("%-52s %5.1f launches  %8.3f G instr  %5.1f%%" % (k[:52], c, v / 1e9, 100 * v / tot))
print("total per proof %.2f G wave-level VALU instructions" % (tot / 1e9))
if len(sys.argv) > 4:
    js = [ln for ln in open(sys.argv[3]).read().splitlines() if ln.startswith("{")]
    b = json.loads(js[-1]) if js else {}
    json.dump({"unit": "wave-level VALU instructions (rocprofv3 --pmc SQ_INSTS_VALU, summed over the launches of one proof)",
               "valu_instructions_per_proof": round(tot), "proofs_profiled": nproofs, "bench_config": b.get("config", {}), "n_gpus": b.get("n_gpus"),
               "kernels": {k: {"launches_per_proof": round(c, 2), "instructions": round(v), "share": round(v / tot, 4)} for v, k, c in sorted(rows, reverse=True)[:8]}},
              open(sys.argv[4], "w"), indent=1)
