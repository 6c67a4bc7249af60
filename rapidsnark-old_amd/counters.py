"""Hardware counters for bench.py's roofline objects: HBM traffic (FETCH_SIZE, WRITE_SIZE) and wave-level VALU instructions
(SQ_INSTS_VALU) of the proof path's kernels.

Two sources, in this order:
  measure()  — rocprofv3 --pmc around a CHILD process (`bench.py --counters-child <legs>`: a few proofs per leg, each beside one more in flight,
               legs separated by marker launches), run by the bench itself after its timed legs: the figures of the line are
               then observed on the box that printed the line.  FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots:
               MI355X_MICROARCH.md, "rocprofv3 PMC slots"); SQ_INSTS_VALU (another block) rides with the first.
  replay()   — the committed passes under profiles/ (same counters, same command, another day's box) when rocprofv3 is
               absent, the run is itself being profiled, or a pass fails: says so in `source`.
Units: FETCH_SIZE / WRITE_SIZE are KiB; raw (uncorrected) bytes are reported — the guide's x2 applies to wide coalesced
streams only, the MSM gathers are random 64 / 128-byte reads.  Measurement plumbing only: nothing here touches the proof path."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK_BEGIN, MARK_END = 16, 80          # marker launches: k_mul_vec<Fr> grids of MARK_BEGIN + leg / MARK_END + leg workgroups (leg = 1, 2, ...)
SETUP_KERNELS = ("precomp", "chain", "build_tables", "pair_tables", "fq_to_internal", "fr_convert", "fr_to_internal", "csr", "fixed_base", "k_mul_vec")


def short_name(kernel_name):
    """'void zk::k_msm_accum_l1<zk::Fp<zk::FqParams> >(args)' -> 'k_msm_accum_l1<Fq>';  G2 kernels get ' [G2]' where only a template says so"""
    n = kernel_name.replace("void ", "").replace("zk::", "").split("(")[0].strip()
    g2 = "Fp2T" in n
    n = n.replace("Fp2T<Fp<FqParams> >", "Fq2").replace("Fp<FqParams>", "Fq").replace("Fp<FrParams>", "Fr").replace(" ", "")
    return n + (" [G2]" if g2 and "g2s" not in n else "")


def available():
    """None when a counter pass can be run from inside this process, else the reason it cannot."""
    if os.environ.get("ZK_BENCH_COUNTERS", "1") == "0":
        return "ZK_BENCH_COUNTERS=0"
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 is not on PATH"
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return "this run is itself under a profiler"
    return None


def run_pass(counters, legs, bench_py, precomp=1, proofs=2, timeout_s=150):
    """One rocprofv3 --pmc pass over the child; -> list of csv rows (dicts).  Raises on any failure."""
    tmp = tempfile.mkdtemp(prefix="zkpmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", ZK_BENCH_COUNTERS="0", GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "16"))
        cmd = ["rocprofv3", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "-o", "c", "--", sys.executable, bench_py,
                                                         "--counters-child", ",".join(legs), "--precomp", str(int(precomp)), "--counters-proofs", str(proofs)]
        r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s, text=True, errors="replace")
        if r.returncode != 0 or "[counters-child] done" not in r.stdout:
            raise RuntimeError("counter pass %s failed (rc %d): %s" % (" ".join(counters), r.returncode, r.stdout[-400:].replace("\n", " | ")))
        rows = []
        for f in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
            rows += list(csv.DictReader(open(f, errors="replace")))
        if not rows:
            raise RuntimeError("counter pass %s wrote no counter_collection.csv" % " ".join(counters))
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cut_legs(rows, legs):
    """rows of one or more passes -> {leg: {kernel short name: {"launches": L, counter: sum over the launches}}} + proofs per leg.
    A leg is what lies between its begin marker and its end marker (the create of the next leg's prover lies outside)."""
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(r)
    out = {leg: {} for leg in legs}
    for counter, rs in by_counter.items():
        rs.sort(key=lambda r: (int(r.get("Process_Id") or 0), int(r["Dispatch_Id"])))
        cur = None
        for r in rs:
            name = r["Kernel_Name"]
            if "k_mul_vec" in name:
                wg = int(r.get("Workgroup_Size") or 256)
                g = int(r.get("Grid_Size") or 0) // max(1, wg)
                if MARK_BEGIN < g <= MARK_BEGIN + len(legs):
                    cur = legs[g - MARK_BEGIN - 1]
                    continue
                if MARK_END < g <= MARK_END + len(legs):
                    cur = None
                    continue
            if cur is None:
                continue
            k = out[cur].setdefault(short_name(name), {"launches": 0})
            k[counter] = k.get(counter, 0.0) + float(r["Counter_Value"])
            k.setdefault("_n_" + counter, 0)
            k["_n_" + counter] += 1
    for leg in out.values():
        for k in leg.values():
            k["launches"] = max([v for kk, v in k.items() if kk.startswith("_n_")] or [0])
            for kk in [kk for kk in k if kk.startswith("_n_")]:
                del k[kk]
    return out


def summarize_leg(kernels):
    """Per-proof / per-launch figures of one leg (see module docstring for units)."""
    proofs = max(1, int(kernels.get("k_spmv_abc", {}).get("launches", 0)))
    work = {k: v for k, v in kernels.items() if not any(x in k for x in SETUP_KERNELS)}

    def hbm(v):
        return (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0

    have_traffic = any("FETCH_SIZE" in v for v in work.values()) and any("WRITE_SIZE" in v for v in work.values())
    have_valu = any("SQ_INSTS_VALU" in v for v in work.values())
    g1 = [v for k, v in work.items() if k.startswith("k_msm_accum_l1<")]
    g2 = [v for k, v in work.items() if k.startswith("k_msm_accum_l1_g2s")]
    ntt = [v for k, v in work.items() if k.startswith("k_ntt_")]
    s = {"proofs_profiled": proofs}
    if have_traffic:
        s["g1_hbm_bytes_per_msm"] = int(sum(hbm(v) for v in g1) / (4 * proofs)) if g1 else None        # four G1 MSMs per proof, whatever the launches
        s["g2_hbm_bytes_per_launch"] = int(sum(hbm(v) for v in g2) / proofs) if g2 else None
        s["transforms_hbm_bytes_per_proof"] = int(sum(hbm(v) for v in ntt) / proofs) if ntt else None
        s["hbm_bytes_per_proof"] = int(sum(hbm(v) for v in work.values()) / proofs)
    if have_valu:
        tot = sum(v.get("SQ_INSTS_VALU", 0.0) for v in work.values())
        s["valu_instructions_per_proof"] = int(tot / proofs)
        top = sorted(((v.get("SQ_INSTS_VALU", 0.0), k) for k, v in work.items()), reverse=True)[:14]
        s["valu_share"] = {k: round(x / tot, 4) for x, k in top if tot}
        s["g1_valu_per_msm"] = int(sum(v.get("SQ_INSTS_VALU", 0.0) for v in g1) / (4 * proofs)) if g1 else None
        s["g2_valu_per_launch"] = int(sum(v.get("SQ_INSTS_VALU", 0.0) for v in g2) / proofs) if g2 else None
        s["transforms_valu_per_proof"] = int(sum(v.get("SQ_INSTS_VALU", 0.0) for v in ntt) / proofs) if ntt else None
    return s


def measure(legs, bench_py, precomp=1, proofs=2, log=None):
    """Both passes -> ({leg: summary}, "measured in this run ...") or raises."""
    rows = run_pass(["FETCH_SIZE", "SQ_INSTS_VALU"], legs, bench_py, precomp, proofs)
    rows += run_pass(["WRITE_SIZE"], legs, bench_py, precomp, proofs)
    cut = cut_legs(rows, legs)
    out = {leg: summarize_leg(k) for leg, k in cut.items()}
    for leg, s in out.items():
        if "valu_instructions_per_proof" not in s or "g1_hbm_bytes_per_msm" not in s:
            raise RuntimeError("counter passes hold no kernels of leg %s" % leg)
    return out, "measured in this run (rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU / --pmc WRITE_SIZE around %d proofs per leg, each submitted beside one more in flight)" % proofs


# ---------------------------------------------------------------- replay of the committed passes (fallback)
def _profile_order(path):
    tag = os.path.basename(path).split("_")[0]
    return (len(tag), tag)


def _same_config(bc, n_gpus, config, world):
    keys = ("log2n", "parallelism", "window_bits", "precomputed_window_tables")
    return all(bc.get(kk) == config.get(kk) for kk in keys) and bc.get("shape", "dense") == config.get("shape", "dense") and n_gpus == world


def replay(config, world):
    """The same summary from profiles/*_counters.json (written by measure() runs that were committed), else from the older
    *_pmc_traffic.json + *_valu_instruction_budget.json pairs; -> (summary | None, source)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), key=_profile_order, reverse=True):
        try:
            d = json.load(open(path))
            for leg in d.get("legs", {}).values():
                if _same_config(leg.get("config", {}), leg.get("n_gpus", 1), config, world):
                    return leg["summary"], "replayed from %s (not measured by this run)" % os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    s, src = {}, []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), key=_profile_order, reverse=True):
        try:
            d = json.load(open(path))
            b = d.get("bench", {})
            if not _same_config(b.get("config", {}), b.get("n_gpus"), config, world):
                continue
            for name, v in d["kernels"].items():
                is_g2 = "k_msm_accum_l1_g2s" in name or ("k_msm_accum_l1" in name and "Fp2T" in name)
                if is_g2:
                    s["g2_hbm_bytes_per_launch"] = int(v["hbm_bytes_raw"])
                elif "k_msm_accum_l1" in name:
                    # that run launched the kernel twice per proof when A, B1, C shared a launch: per MSM = mean per launch x 2 / 4
                    s["g1_hbm_bytes_per_msm"] = int(v["hbm_bytes_raw"] * 2 / 4) if b["config"].get("msm_a_b1_c_in_one_launch") else int(v["hbm_bytes_raw"])
            src.append(os.path.relpath(path, ROOT))
            break
        except (OSError, ValueError, KeyError):
            continue
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_instruction_budget.json")), key=_profile_order, reverse=True):
        try:
            d = json.load(open(path))
            if _same_config(d.get("bench_config", {}), d.get("n_gpus"), config, world):
                s["valu_instructions_per_proof"] = int(d["valu_instructions_per_proof"])
                src.append(os.path.relpath(path, ROOT))
                break
        except (OSError, ValueError, KeyError):
            continue
    if not s:
        return None, "no counter pass for this configuration (not measured, none committed)"
    return s, "replayed from %s (not measured by this run)" % " + ".join(src)


def parse_gather_probe(text):
    """tools/gather_probe's stdout -> [{"row": bytes, "table_mb": MB, "mode": 0|1|2, "bytes_per_s": ...}]
    (mode 0: a lane walks consecutive entries — the level-1 kernels' pattern; 1: lane-interleaved, full occupancy; 2: three waves per SIMD)"""
    out = []
    for m in re.finditer(r"row\s+(\d+) B table\s+(\d+) MB mode (\d): +([\d.]+) ms +([\d.]+) GB/s", text):
        out.append({"row": int(m.group(1)), "table_mb": int(m.group(2)), "mode": int(m.group(3)), "bytes_per_s": float(m.group(5)) * 1e9})
    return out


def gather_ceiling(row_bytes, footprint_bytes):
    """The chip's rate for random bursts of `row_bytes` out of a table of the launch's footprint, from the newest committed output
    of tools/gather_probe (profiles/*_gather_probe.txt): the best of the probe's access patterns at the smallest measured table
    that is at least as large as the footprint (the rate falls with the table: 3.5 GB -> 7 GB halves it).  -> dict | None"""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gather_probe.txt")), key=_profile_order, reverse=True):
        try:
            rows = [r for r in parse_gather_probe(open(path).read()) if r["row"] == row_bytes]
        except OSError:
            continue
        if not rows:
            continue
        sizes = sorted({r["table_mb"] for r in rows})
        want = footprint_bytes / float(1 << 20)
        mb = next((x for x in sizes if x >= want * 0.93), sizes[-1])
        best = max((r for r in rows if r["table_mb"] == mb), key=lambda r: r["bytes_per_s"])
        return {"bytes_per_s": best["bytes_per_s"], "table_mb": mb, "mode": best["mode"], "source": os.path.relpath(path, ROOT)}
    return None
