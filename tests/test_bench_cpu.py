"""bench.py's measurement plumbing that needs no GPU: how the counter passes the bench runs around a child of itself are cut
into legs and summarised (rapidsnark_old_amd.counters), what is replayed from the committed passes when they cannot be run, how
the roofline's counter-derived fields follow from them, and the tool that cuts a kernel trace of the driver's command into one
table per leg (profiles/INDEX.md says how every figure of the line is recomputed from those tables)."""
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


def test_counters_are_replayed_from_the_committed_passes_when_they_cannot_be_measured():
    from rapidsnark_old_amd import counters
    config, _ = _config()
    cs, src = counters.replay(config, 1)
    assert "replayed from" in src and "not measured by this run" in src
    assert 13.0e9 < cs["valu_instructions_per_proof"] < 16.0e9 and 4.0e9 < cs["g1_hbm_bytes_per_msm"] < 5.5e9 and 3.5e9 < cs["g2_hbm_bytes_per_launch"] < 5.0e9
    # another configuration has no pass of its own: nothing is replayed, nothing is invented
    none, why = counters.replay(dict(config, log2n=19, window_bits=17), 1)
    assert none is None and "no counter pass" in why


def test_roofline_fields_follow_from_the_counters_and_the_sampled_clock():
    out = {"ms_per_step": 32.7, "roofline": {"algorithmic_bytes": 402653184, "launch_ms_one_in_flight": 3.9, "simds": 1024, "clock_ghz": 2.0, "gather_table_bytes": 3 * 13 * (1 << 22) * 64}}
    cs = {"g1_hbm_bytes_per_msm": 4680000000, "g2_hbm_bytes_per_launch": 4200000000, "valu_instructions_per_proof": 15400000000}
    bench.finish_roofline(out, cs, "measured in this run (test)", {"bytes_per_s": 1.37e12, "table_mb": 10240, "mode": 1, "source": "profiles/x_gather_probe.txt"})
    r = out["roofline"]
    assert r["traffic"] == 4680000000 and abs(r["traffic_ratio"] - 11.623) < 0.01
    assert abs(r["gather_bytes_per_s"] - 1.2e12) < 1e9 and abs(r["gather_frac"] - 0.8759) < 1e-3
    want = 15.4e9 * 4.0 / (1024 * 2.0e9) * 1e3
    assert abs(r["issue_bound_ms"] - want) < 1e-2 and abs(r["issue_frac"] - want / 32.7) < 1e-3
    # no telemetry on the box: the instructions are still reported, the bound is not priced at a guessed clock
    out2 = {"ms_per_step": 32.7, "roofline": {"algorithmic_bytes": 402653184, "launch_ms_one_in_flight": None, "simds": 1024, "clock_ghz": None}}
    bench.finish_roofline(out2, cs, "x", None)
    assert out2["roofline"]["valu_instructions_per_proof"] and out2["roofline"]["issue_bound_ms"] is None and out2["roofline"]["gather_frac"] is None
    # no counters at all
    out3 = {"ms_per_step": 9.0, "roofline": {"algorithmic_bytes": 100663296, "launch_ms_one_in_flight": 1.0, "simds": 1024, "clock_ghz": 2.0}}
    bench.finish_roofline(out3, None, "no counter pass for this configuration", None)
    assert out3["roofline"]["traffic"] is None and out3["roofline"]["issue_frac"] is None


def test_counter_rows_are_cut_into_legs_at_the_marker_launches():
    """what rapidsnark_old_amd.counters makes of rocprofv3's counter_collection.csv rows (two passes, two legs)"""
    from rapidsnark_old_amd import counters
    rows, did = [], [0]

    def k(name, counter, value, grid=256 * 64, wg=256):
        did[0] += 1
        rows.append({"Dispatch_Id": str(did[0]), "Process_Id": "7", "Kernel_Name": name, "Grid_Size": str(grid), "Workgroup_Size": str(wg),
                     "Counter_Name": counter, "Counter_Value": str(float(value))})

    mul = "void zk::k_mul_vec<zk::Fp<zk::FrParams> >(a, b, c, n)"
    l1 = "void zk::k_msm_accum_l1<zk::Fp<zk::FqParams> >(x)"
    g2 = "zk::k_msm_accum_l1_g2s(x)"
    for counter, per_l1, per_g2 in (("FETCH_SIZE", 9000000.0, 4000000.0), ("SQ_INSTS_VALU", 3.5e9, 5.6e9), ("WRITE_SIZE", 100000.0, 50000.0)):
        did[0] = 0                                                   # (each pass is its own process: dispatch ids start over)
        k("void zk::k_precomp_walk<zk::Fp<zk::FqParams> >(t)", counter, 1e12)      # create of leg 1: outside every leg
        k(mul, counter, 1, grid=256 * (counters.MARK_BEGIN + 1))
        for _ in range(3):                                           # three proofs: A|B1|C in one launch + H, one G2 launch
            k("zk::k_spmv_abc(y)", counter, 10)
            k(l1, counter, per_l1); k(l1, counter, per_l1); k(g2, counter, per_g2)
            k("void zk::k_ntt_mid<0>(z)", counter, 7)
        k(mul, counter, 1, grid=256 * (counters.MARK_END + 1))
        k("void zk::k_precomp_walk<zk::Fp<zk::FqParams> >(t)", counter, 1e12)      # create of leg 2
        k(mul, counter, 1, grid=256 * (counters.MARK_BEGIN + 2))
        k("zk::k_spmv_abc(y)", counter, 10); k(l1, counter, 5 * per_l1)
        k(mul, counter, 1, grid=256 * (counters.MARK_END + 2))
        k(mul, counter, 1, grid=256 * 4)                             # an ordinary operator call after the legs
    cut = counters.cut_legs(rows, ["2p22", "2p20"])
    assert set(cut["2p22"]) == {"k_spmv_abc", "k_msm_accum_l1<Fq>", "k_msm_accum_l1_g2s", "k_ntt_mid<0>"} and cut["2p22"]["k_msm_accum_l1<Fq>"]["launches"] == 6
    s = counters.summarize_leg(cut["2p22"])
    assert s["proofs_profiled"] == 3
    assert s["g1_hbm_bytes_per_msm"] == int((9000000.0 + 100000.0) * 1024 * 2 / 4)          # two launches carry four MSMs
    assert s["g2_hbm_bytes_per_launch"] == int((4000000.0 + 50000.0) * 1024)
    assert s["valu_instructions_per_proof"] == int(2 * 3.5e9 + 5.6e9 + 10 + 7) and s["g1_valu_per_msm"] == int(3.5e9 / 2)
    s2 = counters.summarize_leg(cut["2p20"])
    assert s2["proofs_profiled"] == 1 and s2["g1_hbm_bytes_per_msm"] == int(5 * 9100000.0 * 1024 / 4)
    assert counters.short_name("void zk::k_msm_reduce_chunks<zk::Fp2T<zk::Fp<zk::FqParams> > >(a)") == "k_msm_reduce_chunks<Fq2> [G2]"


def test_gather_probe_output_is_parsed_and_the_ceiling_follows_the_table_size(tmp_path, monkeypatch):
    from rapidsnark_old_amd import counters
    txt = "row  64 B table  3584 MB mode 0:   1.912 ms  1825.4 GB/s   28.52 G rows/s\nrow  64 B table  3584 MB mode 2:   1.130 ms  3086.9 GB/s   48.23 G rows/s\n" \
          "row  64 B table  7168 MB mode 1:   2.576 ms  1354.5 GB/s   21.16 G rows/s\nrow 128 B table  7168 MB mode 1:   5.044 ms  1383.8 GB/s   10.81 G rows/s\n"
    rows = counters.parse_gather_probe(txt)
    assert len(rows) == 4 and rows[1] == {"row": 64, "table_mb": 3584, "mode": 2, "bytes_per_s": 3086.9e9}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r09a_gather_probe.txt").write_text(txt)
    monkeypatch.setattr(counters, "ROOT", str(tmp_path))
    one = counters.gather_ceiling(64, 13 * (1 << 22) * 64)              # one G1 table at 2^22: 3328 MB -> the 3584 MB measurement, best pattern
    assert one["table_mb"] == 3584 and one["mode"] == 2 and abs(one["bytes_per_s"] - 3086.9e9) < 1
    three = counters.gather_ceiling(64, 3 * 13 * (1 << 22) * 64)        # A|B1|C in one launch: 9984 MB -> the largest table measured
    assert three["table_mb"] == 7168 and abs(three["bytes_per_s"] - 1354.5e9) < 1
    assert counters.gather_ceiling(32, 1 << 30) is None


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


def test_summary_carries_the_2p24_leg_and_the_rank_share_of_eight():
    """The scaling evidence of a one-GPU run (also_2p24, also_shard8) must be among the flat scalars at the END of the line: a record
    that keeps only the tail of stdout still holds them.  A probe that failed is reported as such and adds no scalar."""
    out = {"value": 32.0, "ms_per_step": 31.2, "roofline": {"frac": 0.008, "launch_ms": 6.2, "g2_launch_ms": 16.0, "whole_proof_frac": 0.024},
           "also_2p24": {"value": 8.5, "ms_per_step": 117.7, "ms_per_proof_sync": 133.1},
           "also_shard8": {"2p22": {"log2n": 22, "rank_share_ms_one_at_a_time": 5.8, "rank_share_ms_two_in_flight": 5.0, "ideal_share_ms": 3.9,
                                    "implied_speedup_two_in_flight": 6.24, "kernel_launches_per_rank": 101, "additions_per_point_h": 15},
                           "2p24": {"log2n": 24, "error": "HipError: out of memory"}}}
    s = bench.summary_of(out)
    assert s["ms_per_step_2p24"] == 117.7 and s["proofs_per_s_2p24"] == 8.5 and s["ms_per_proof_sync_2p24"] == 133.1
    assert s["shard8_2p22_rank_ms_two_in_flight"] == 5.0 and s["shard8_2p22_ideal_ms"] == 3.9 and s["shard8_2p22_implied_speedup"] == 6.24
    assert s["shard8_2p22_launches_per_rank"] == 101 and s["shard8_2p22_additions_per_point"] == 15
    assert not any(kk.startswith("shard8_2p24") for kk in s)
    out["also_2p24"] = {"value": None, "error": "x"}
    assert "ms_per_step_2p24" not in bench.summary_of(out)
