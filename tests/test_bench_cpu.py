"""bench.py's measurement plumbing that needs no GPU: what the driver-recorded line REPLAYS from the committed counter passes
(roofline.traffic, roofline.issue_bound), the mirror of the library's launch rules, and the tool that cuts a kernel trace of the
driver's command into one table per leg (profiles/INDEX.md says how every figure of the line is recomputed from those tables)."""
import csv
import json
import os
import subprocess
import sys
import types

from conftest import ROOT

import bench


def _config():
    d = json.loads([ln for ln in open(os.path.join(ROOT, "profiles", "r04z_bench.json")) if ln.startswith("{")][-1])
    return d["config"], d


def test_issue_bound_is_replayed_from_the_committed_counter_pass():
    config, line = _config()
    clock = {"clock_ghz": 2.0, "power_w": 1100.0, "samples": 20, "source": "test"}
    ib = bench.issue_bound_from_profiles(config, 1, 256, clock, 32.7)
    assert ib["valu_instructions_per_proof"] > 15.0e9 and "valu_instruction_budget.json" in ib["instructions_source"]
    assert ib["simds"] == 1024 and ib["cycles_per_instruction"] == 4.0
    want = ib["valu_instructions_per_proof"] * 4.0 / (1024 * 2.0e9) * 1e3
    assert abs(ib["bound_ms"] - want) < 1e-2 and abs(ib["achieved_frac"] - want / 32.7) < 1e-3
    # another configuration (2^20) has no counter pass of its own: nothing is replayed, nothing is invented
    other = dict(config, log2n=20, window_bits=19)
    ib20 = bench.issue_bound_from_profiles(other, 1, 256, clock, 9.3)
    assert ib20["valu_instructions_per_proof"] is None and ib20["bound_ms"] is None
    # no telemetry on the box: the instructions are still reported, the bound is not priced at a guessed clock
    ib0 = bench.issue_bound_from_profiles(config, 1, 256, {"clock_ghz": None, "source": "no telemetry on this box"}, 32.7)
    assert ib0["valu_instructions_per_proof"] and ib0["bound_ms"] is None
    # the committed line itself is consistent with its own inputs
    li = line["roofline"]["issue_bound"]
    assert abs(li["bound_ms"] - li["valu_instructions_per_proof"] * 4.0 / (li["simds"] * li["clock_ghz"] * 1e9) * 1e3) < 0.01
    assert abs(li["achieved_frac"] - li["bound_ms"] / line["ms_per_step"]) < 1e-3


def test_traffic_is_per_msm_when_a_b1_c_share_a_launch():
    config, _ = _config()
    args = types.SimpleNamespace(traffic_bytes=None)
    g1, src = bench.traffic_from_profiles(args, config, 1, "g1")
    g2, _ = bench.traffic_from_profiles(args, config, 1, "g2")
    import re
    raw = json.load(open(os.path.join(ROOT, re.search(r"profiles/\w+_pmc_traffic\.json", src).group(0))))      # the file the line says it replays
    per_launch = [v["hbm_bytes_raw"] for k, v in raw["kernels"].items() if "k_msm_accum_l1<" in k][0]
    assert raw["bench"]["config"]["msm_a_b1_c_in_one_launch"] is True
    assert g1 == int(per_launch * 2 / 4) and "per G1 MSM" in src and 4.0e9 < g1 < 5.5e9          # two launches carry four MSMs
    assert 3.5e9 < g2 < 5.0e9
    assert bench.traffic_from_profiles(args, dict(config, log2n=19), 1, "g1")[0] is None


def test_batch_rule_mirrors_the_library():
    os.environ.pop("ZKHIP_BATCH_ABC", None)
    assert bench.batch_abc_default(1 << 19, 1) and bench.batch_abc_default(1 << 22, 1) and bench.batch_abc_default(1 << 24, 1)
    assert not bench.batch_abc_default(1 << 20, 1) and not bench.batch_abc_default((1 << 22) - 1, 1)
    assert bench.batch_abc_default(1 << 20, 4) and bench.batch_abc_default(1 << 21, 8)                  # shards: always
    src = open(os.path.join(ROOT, "rapidsnark-old_amd", "csrc", "prover.hip")).read()
    assert "p->shard_count == 1 && p->sv.size() >= (1u << 20) && p->sv.size() < (1u << 22)" in src      # the rule it mirrors


def test_leg_stats_cuts_a_trace_at_the_marker_launches(tmp_path):
    d = tmp_path / "trace"
    d.mkdir()
    cols = ["Kind", "Agent_Id", "Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X"]
    rows, t = [], [1000]

    def k(name, dur, grid=256 * 64):
        rows.append(["KERNEL_DISPATCH", 1, 1, name, t[0], t[0] + dur, 256, grid])
        t[0] += dur + 10

    k("void zk::k_setup()", 500)
    k("void zk::k_mul_vec<zk::Fp<zk::FrParams> >(a, b)", 5, grid=256 * 17)          # marker of leg 1
    for _ in range(3):
        k("void zk::k_msm_accum_l1<zk::Fp<zk::FqParams> >(x)", 4000)
        k("zk::k_spmv_abc(y)", 500)
    k("void zk::k_mul_vec<zk::Fp<zk::FrParams> >(a, b)", 5, grid=256 * 18)          # marker of leg 2
    k("void zk::k_msm_accum_l1<zk::Fp<zk::FqParams> >(x)", 3000)
    k("void zk::k_mul_vec<zk::Fp<zk::FrParams> >(a, b)", 7, grid=256 * 4)           # an ordinary operator call: not a marker
    with open(d / "s_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        w.writerows(rows)
    err = tmp_path / "err.txt"
    err.write_text("[bench] leg marker 1 (grid of 17 workgroups): 2p22_headline\nnoise\n[bench] leg marker 2 (grid of 18 workgroups): 2p22_lone_resident\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "leg_stats.py"), str(d), str(err), str(tmp_path / "t_kernel_stats")],
                         capture_output=True, text=True, check=True).stdout
    assert "2p22_headline" in out and "2p22_lone_resident" in out and "before_first_marker" in out
    head = list(csv.DictReader(open(tmp_path / "t_kernel_stats_2p22_headline.csv")))
    l1 = [r for r in head if "k_msm_accum_l1" in r["Name"]][0]
    assert int(l1["Calls"]) == 3 and float(l1["AverageNs"]) == 4000.0 and int(float(l1["TotalDurationNs"])) == 12000
    lone = list(csv.DictReader(open(tmp_path / "t_kernel_stats_2p22_lone_resident.csv")))
    assert {r["Name"].split("<")[0].split("::")[-1].split("(")[0] for r in lone} == {"k_msm_accum_l1", "k_mul_vec"}     # the small k_mul_vec stays a kernel of the leg
