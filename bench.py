#!/usr/bin/env python
"""bench.py — Groth16 prove() throughput on MI355X (BASELINE.json metric: proofs/s and ms/proof at 2^20 and 2^22).

A "step" is one whole proof (reference src/groth16.cpp:48-254) of a synthetic BN254 key with n = domainSize = nVars = 2^k
(default k = 22: BASELINE configs[2]; --log2n 20 = configs[1]).  Every step has its own witness in pageable HOST memory, as
the reference's Prover::prove(FrElement *wtns) contract has it (src/groth16.hpp:101): the upload is INSIDE the timed region
(zk_prove_submit).  The rate with witnesses resident in HBM is measured too (`resident_witness`).

  python bench.py --gpus 1 --steps K --warmup W                                  (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1: one proof split across the ranks,
      MSM tables sharded by point range, the chain partitioned, four RCCL all_to_all per proof; "scaling": "strong")

ONE JSON line on rank 0.  Besides the contract's keys:
  roofline      flat: the dominant kernel k_msm_accum_l1<Fq> against HBM (algorithmic bytes 96*n per G1 MSM / launch time, SURVEY
                section 8d), its measured HBM traffic and random-gather rate against the chip's gather ceiling, the G2 launch (g2_*),
                and the path's own bound — VALU issue (issue_*: instructions per proof x 4 cycles / (SIMDs x sampled clock)).
                Counters come from rocprofv3 passes the bench runs itself after the timed legs (rapidsnark_old_amd.counters),
                or are replayed from profiles/ when it cannot: `traffic_source` says which.
  cpu_baseline  the C restatement of rapidsnark's CPU algorithm (oracle/) timed on this box's host cores, generic and ADX builds.
  also_2p20, also_realistic, (also_server)   the other configurations of BASELINE's metric, timed by the same code.
  also_2p24     configs[3]'s size on ONE GPU (pipelined period, synchronous latency), same code.
  also_shard8   for 2^22 and 2^24: ONE rank's share of a proof split across eight GPUs (MSM tables sharded by point range, chain
                partitioned), measured on this GPU with the four all_to_all rounds left out; beside it the ideal eighth of this
                run's single-GPU period and the speed-up the share implies.  No number from eight physical GPUs: the driver
                owns those runs (SCALE_rNN.json).
  summary       LAST in the line: every scalar DESIGN.md quotes, so that a record keeping only the tail of stdout holds them.
Definitions of every field: DESIGN.md section 6.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ZK_DEPTH_MAX = 8               # ZK_MAX_IN_FLIGHT (include/zkhip.h)
G1_MSM_BYTES_PER_POINT = 96    # 64 B affine point + 32 B scalar
G2_MSM_BYTES_PER_POINT = 160   # 128 B affine G2 point + 32 B scalar


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=22)
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--witness", choices=["uniform", "realistic"], default="uniform",
                    help="uniform = BASELINE's random witness (worst case); realistic = 80%% {0,1}, 15%% < 2^32, 5%% full")
    ap.add_argument("--shape", choices=["dense", "circuit"], default="dense",
                    help="dense = BASELINE's family (nVars = domainSize, every table row a point); circuit = nVars = 3/4 of the domain + 5, "
                         "3 public signals, ~30%% of the rows of A and of B1/B2 at infinity (rapidsnark_old_amd.synth)")
    ap.add_argument("--sparse-witness", type=int, default=-1,
                    help="1 = ZK_FLAG_SPARSE_WITNESS: 16-bit window for the four witness MSMs (what a deployment for circom circuits sets); "
                         "default: 1 with --witness realistic, else 0")
    ap.add_argument("--no-realistic", action="store_true", help="skip the also_realistic leg of a default (2^22, N = 1) run (--no-cpu skips it too)")
    ap.add_argument("--precomp", type=int, default=1,
                    help="1 (default) = window-precomputed point tables resident in HBM (ZK_FLAG_PRECOMP: one-off work in create, like\n"
                         "the reference's makeProver it is outside the timed prove()); 0 = tables exactly as in the zkey;\n"
                         "2 = rows for every second window (ZK_FLAG_PRECOMP_HALF: 7 instead of 13 x the table memory, two bucket reductions per MSM)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default, single GPU): consecutive proofs overlap through zk_prove_dev_submit / zk_prove_collect "
                         "(two in flight); 0: strictly one proof at a time (zk_prove_dev)")
    ap.add_argument("--in-flight", type=int, default=0, help="proofs in flight per GPU (0 = by size: 8 below 2^19, 6 up to 2^22, above 3 with host witnesses / 2 with resident ones; max 8)")
    ap.add_argument("--batch", type=int, default=0, help="N = 1, host witnesses: B >= 2 = B witnesses per submission (zk_prove_batch_submit; small circuits). --steps must be a multiple of B")
    ap.add_argument("--collector-thread", type=int, default=1, help="N = 1: collect on a second host thread (1, default) or in the submitting thread (0)")
    ap.add_argument("--witness-in", choices=["host", "hbm"], default="host",
                    help="host (default): witnesses are pageable host arrays and every upload is timed (the reference's contract); "
                         "hbm: witnesses resident in HBM before the timed region")
    ap.add_argument("--chain", choices=["auto", "partitioned", "replicated"], default="auto",
                    help="N > 1: partition the A.w/B.w rows and the NTTs across the ranks (auto: when N is 2, 4 or 8) or replicate them")
    ap.add_argument("--verify", type=int, default=1, help="N > 1: check one sharded proof against an unsharded prover on rank 0 (outside the timed region)")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the `replicas` leg (every GPU proving its own proofs, no collective)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-2p20", action="store_true", help="skip the also_2p20 leg of a default (2^22, N = 1) run (--no-cpu skips it too)")
    ap.add_argument("--cpu-budget-s", type=float, default=36.0)
    ap.add_argument("--no-counters", action="store_true", help="do not run the rocprofv3 counter passes after the timed legs (replay profiles/ instead)")
    ap.add_argument("--no-2p24", action="store_true", help="skip the also_2p24 leg of a default run (configs[3]'s size on one GPU, and its rank share of eight)")
    ap.add_argument("--no-shard8", action="store_true", help="skip the also_shard8 probes of a default run (one rank's share of a proof split across eight GPUs)")
    ap.add_argument("--no-server", action="store_true", help="skip the also_server leg of a default run (proverServer over REST on a Semaphore-class key)")
    ap.add_argument("--counters-child", default="", help=argparse.SUPPRESS)      # internal: the process rocprofv3 wraps (rapidsnark_old_amd.counters)
    ap.add_argument("--counters-proofs", type=int, default=2, help=argparse.SUPPRESS)
    return ap.parse_args()


LEG_OF = {("dense", 22): "2p22", ("circuit", 22): "2p22_circuit", ("dense", 20): "2p20"}      # legs of a default run that get counters


def main():
    args = parse()
    if args.counters_child:
        counters_child(args)
        return
    out = run(args)
    if out is None:                       # ranks > 0
        return
    default_run = args.gpus == 1 and args.log2n == 22 and not args.no_cpu and args.batch <= 1 and args.witness == "uniform" and args.shape == "dense"
    legs = {LEG_OF.get((args.shape, args.log2n), "headline"): out}
    import copy
    shard8 = []
    if "shard8" in out:
        shard8.append(out.pop("shard8"))
    if default_run and not args.no_realistic:
        # what a REAL circom key and witness look like to the prover (BASELINE configs[4]'s fidelity; SURVEY section 8d's
        # secondary line): the circuit-shaped member of the family with the 80/15/5 witness, timed the same way by the same code
        a3 = copy.copy(args)
        a3.witness, a3.shape, a3.no_cpu, a3.in_flight = "realistic", "circuit", True, 0
        legs["2p22_circuit"] = run(a3)
    if default_run and not args.no_2p20:
        # BASELINE's metric is quoted "at 2^20 and 2^22": configs[1], timed the same way by the same code
        a2 = copy.copy(args)
        a2.log2n, a2.no_cpu, a2.in_flight = 20, True, 0
        legs["2p20"] = run(a2)
    also_2p24 = None
    if default_run and not args.no_2p24:
        # BASELINE configs[3]'s size on ONE GPU (it is "sharded across 8" there: the driver owns that run), and — on the same
        # workload, before it is released — one rank's share of eight
        a4 = copy.copy(args)
        a4.log2n, a4.no_cpu, a4.in_flight = 24, True, 0
        a4.steps, a4.warmup = min(args.steps, 8), min(args.warmup, 2)
        try:
            o4 = run(a4)
            if "shard8" in o4:
                shard8.append(o4.pop("shard8"))
            also_2p24 = {kk: o4[kk] for kk in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_per_proof_sync", "config", "resident_witness",
                                               "latency_ms_one_at_a_time", "stage_ms", "setup_s") if kk in o4}
            r4 = o4["roofline"]
            also_2p24["roofline"] = {kk: r4[kk] for kk in ("kernel", "achieved", "peak", "unit", "frac", "launch_ms", "algorithmic_bytes", "launch_ms_one_in_flight",
                                                          "frac_one_in_flight", "g2_launch_ms", "g2_launch_ms_one_in_flight", "whole_proof_frac", "clock_ghz") if kk in r4}
        except Exception as exc:           # noqa: BLE001  (a failed extra leg must not cost the line)
            also_2p24 = {"value": None, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    # ---- counters: measured by this run where rocprofv3 can wrap a child of it, else replayed from profiles/
    from rapidsnark_old_amd import counters
    measured, how = {}, None
    why_not = counters.available() if not args.no_counters else "--no-counters"
    can = [leg for leg in legs if leg in LEG_OF.values()]
    if why_not is None and args.gpus == 1 and can:
        try:
            measured, how = counters.measure(can, os.path.abspath(__file__), precomp=args.precomp, proofs=args.counters_proofs)
        except Exception as exc:           # noqa: BLE001  (a failed pass must not cost the line)
            why_not = "%s: %s" % (type(exc).__name__, str(exc)[:200])
    for leg, o in legs.items():
        if leg in measured:
            cs, src = measured[leg], how
        else:
            cs, src = counters.replay(o["config"], o["n_gpus"])
            if why_not:
                src += "; in-run passes skipped: " + why_not
        finish_roofline(o, cs, src, counters.gather_ceiling(64, o["roofline"]["gather_table_bytes"]))
    tag = os.environ.get("ZK_BENCH_SAVE_COUNTERS")          # tools/profile_bench.sh: keep this run's counter summaries for later replays
    if tag and measured:
        with open(os.path.join(ROOT, "profiles", "%s_counters.json" % tag), "w") as f:
            json.dump({"source": how, "legs": {leg: {"config": legs[leg]["config"], "n_gpus": legs[leg]["n_gpus"], "summary": measured[leg]} for leg in measured}}, f, indent=1)
    for leg, key in (("2p22_circuit", "also_realistic"), ("2p20", "also_2p20")):
        if leg in legs and legs[leg] is not out:
            o = legs[leg]
            out[key] = {kk: o[kk] for kk in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_per_proof_sync", "config", "roofline", "resident_witness",
                                             "latency_ms_one_at_a_time", "stage_ms") if kk in o}
    if also_2p24 is not None:
        out["also_2p24"] = also_2p24
    if shard8:
        out["also_shard8"] = {"2p%d" % o["log2n"]: o for o in shard8}
    if default_run and not args.no_server:
        out["also_server"] = server_leg()
    out["summary"] = summary_of(out)            # LAST: the scalars DESIGN.md quotes, within the tail of the line
    print(json.dumps(out), flush=True)


def run(args):
    if args.batch > 1 and (args.steps % args.batch or args.gpus != 1):
        raise SystemExit("--batch B needs --gpus 1 and --steps a multiple of B")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # six streams per prover (csrc/prover_create.hip); read when HIP initialises
    import torch
    import rapidsnark_old_amd as zk
    from rapidsnark_old_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Test hook (single-GPU boxes): ZK_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and exchanges
    # over gloo, so the whole N>1 path except the RCCL call itself can be exercised on one GPU.
    share = os.environ.get("ZK_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = torch.device("cpu") if share else dev          # where the 384-byte records are exchanged
    dist = None
    first_contact_info = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        first_contact_info = multi_gpu_first_contact(args, dist, torch, dev, xdev, rank, world, share)

    k = args.log2n
    n = 1 << k
    t0 = time.time()
    # --- synthetic zkey (tables generated by the product's GPU chain kernels)
    wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes(), shape=args.shape)
    t_gen = time.time() - t0
    t0 = time.time()
    partitioned = world in (2, 4, 8) and args.chain != "replicated" and k >= 6
    if args.chain == "partitioned" and not partitioned:
        raise SystemExit("--chain partitioned needs 2, 4 or 8 ranks")
    sparse = bool(args.sparse_witness) if args.sparse_witness >= 0 else (args.witness == "realistic" and bool(args.precomp) and not args.window_bits)
    prover = ProverFromView(zk, wl, device=local_rank, shard_index=rank, shard_count=world,
                            window_bits=args.window_bits, timings=True, precomp=args.precomp, partitioned_chain=partitioned,
                            batch=args.batch if world == 1 else 0, sparse_witness=sparse)
    t_create = time.time() - t0
    plan = prover.info()                    # zk_prover_info: what the library decided (windows, A|B1|C in one launch, lanes, depths)

    def depth_for(in_hbm):
        return args.in_flight or plan["depth_resident_witness" if in_hbm else "depth_host_witness"]

    chain = None
    sliced_upload = False
    if partitioned:
        from rapidsnark_old_amd.dist import ShardedChain

        def exchange_via_cpu(dst, src):
            # single-GPU test hook only (gloo): all_gather of the send buffers on the host, then pick this rank's parts
            torch.cuda.synchronize()
            mine = src.cpu()
            allb = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allb, mine)
            part = mine.numel() // world
            dst.copy_(torch.cat([allb[sidx][rank * part:(rank + 1) * part] for sidx in range(world)]))
            torch.cuda.synchronize()

        chain = ShardedChain(prover.lib, prover.h, dist, dev, exchange=exchange_via_cpu if share else None)

        def gather_via_cpu(full, part):                   # single-GPU test hook only (gloo)
            torch.cuda.synchronize()
            mine = part.cpu()
            allp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allp, mine)
            full.copy_(torch.cat(allp))
            torch.cuda.synchronize()

        sliced_upload = chain.enable_sliced_upload(wl["nVars"], rank, world, ZK_DEPTH_MAX, dev, gather=gather_via_cpu if share else None)

    # --- distinct witnesses (same on every rank: seeded): pageable host arrays, and HBM copies of
    # the same for the resident-witness leg
    nw = args.steps + args.warmup
    n_wits = max(1, min(nw, (8 << 30) // (32 << k)))      # one per step while they fit 8 GiB of host memory (2^22: all 33), else cycled
    wits_host = [synth.make_witness(k, seed=i, kind=args.witness, n_vars=wl["nVars"]) for i in range(n_wits)]
    wits_dev = [torch.from_numpy(w).to(dev) for w in wits_host]
    torch.cuda.synchronize()
    pipelined = bool(args.pipeline)

    leg_prefix = "2p%d%s" % (args.log2n, "" if args.shape == "dense" else "_" + args.shape)

    def timed_run(in_hbm, steps, warmup, leg=None):
        """`warmup` untimed + exactly `steps` timed whole proofs, each with its own witness and random
        r,s (like the reference).  -> (seconds, mean stage timings).  leg: name of the profile leg the TIMED region
        forms (a marker launch in front of it when ZK_BENCH_LEG_MARKERS=1; nothing in a normal run)."""
        units = args.batch if (args.batch > 1 and world == 1 and not in_hbm) else 1     # proofs per submission
        nsub = steps // units

        def submit(i):
            j = i % len(wits_host)
            if units > 1:                                    # zk_prove_batch_submit: `units` witnesses, one set of launches
                prover.submit_batch([wits_host[(i * units + t) % len(wits_host)] for t in range(units)])
                return
            if chain is not None:                            # phases + four rounds of all_to_all (RCCL over xGMI)
                if in_hbm:
                    chain.submit(d_wtns=wits_dev[j].data_ptr())
                elif sliced_upload:                          # 1/N of the witness over PCIe per rank + all_gather over xGMI
                    chain.submit_host_sliced(wits_host[j])
                else:
                    chain.submit(wtns=wits_host[j])
            elif in_hbm:
                prover.submit_dev(wits_dev[j].data_ptr())
            else:
                prover.submit_host(wits_host[j])             # pageable -> pinned staging -> HBM, all inside the call / its stream

        def collect():
            if units > 1:
                prover.collect_batch(units)
            elif world == 1:
                prover.collect()
            else:                                            # this rank's partial sums -> all ranks -> rank 0 assembles
                parts = zk.gather_partials(prover.collect_msm(), dist, xdev)     # RCCL all_gather over xGMI: 384 B per rank
                if rank == 0:
                    prover.prove_finish(parts)

        stage = {}

        def add_timings():
            for kk, v in prover.timings().items():
                stage[kk] = stage.get(kk, 0.0) + v

        for i in range(warmup):
            submit(i)
            collect()
        if pipelined and warmup:
            # still untimed: fill the pipeline once, so that every proof slot the timed loop uses exists (a slot's
            # buffers are allocated the first time its depth is reached)
            depth0 = depth_for(in_hbm)
            for i in range(depth0):
                submit(i)
            for i in range(depth0):
                collect()
        if leg:
            leg_marker(zk, torch, leg)           # behind the warm-up and the pipeline fill: the leg holds the timed proofs only
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pipelined:
            # up to `depth` proofs in flight: proof i+1 is enqueued (witness upload included) before proof
            # i is collected, so its sort, SpMV and NTTs overlap proof i's reductions, D2H and host tail;
            # with host witnesses a third proof hides the upload (a proof cannot start before its witness
            # has arrived, and a slot is only free again after a collect)
            depth = depth_for(in_hbm)
            if world == 1 and args.collector_thread:
                # two host threads, as a server would run them: this one submits, the other collects (wait for
                # the GPU + 0.3 ms of host tail per proof: window sums, final assembly).  The
                # library holds its submission mutex only around bookkeeping during a collect; ctypes drops
                # the GIL for both calls.
                import threading
                free = threading.Semaphore(depth)
                ready = threading.Semaphore(0)            # proofs submitted and not yet collected
                errors = []

                def collector():
                    try:
                        for _ in range(nsub):
                            ready.acquire()               # never ask for a proof that has not been submitted yet
                            if errors:
                                return
                            collect()
                            add_timings()
                            free.release()
                    except Exception as exc:              # noqa: BLE001
                        errors.append(exc)
                        for _ in range(nsub):
                            free.release()

                th = threading.Thread(target=collector)
                started = False
                for i in range(nsub):
                    free.acquire()
                    if errors:
                        break
                    try:
                        submit(warmup + i)
                    except Exception as exc:              # noqa: BLE001
                        errors.append(exc)
                        for _ in range(nsub):
                            ready.release()
                        break
                    ready.release()
                    if not started:
                        th.start()
                        started = True
                if started:
                    th.join()
                if errors:
                    raise errors[0]
            else:
                flying = 0
                for i in range(nsub):
                    submit(warmup + i)
                    flying += 1
                    if flying == depth:
                        collect()
                        add_timings()
                        flying -= 1
                while flying:
                    collect()
                    add_timings()
                    flying -= 1
        else:
            for i in range(nsub):
                submit(warmup + i)
                collect()
                add_timings()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, {kk: v / nsub for kk, v in stage.items()}

    headline_hbm = args.witness_in == "hbm"
    from rapidsnark_old_amd.telemetry import ClockSampler
    leg_marker(zk, torch, leg_prefix + "_warmup")
    with ClockSampler(local_rank, enabled=os.environ.get("ZK_BENCH_CLOCK", "1") != "0") as clock:   # shader clock + socket power WHILE the headline runs (issue-bound roofline)
        elapsed, stage = timed_run(headline_hbm, args.steps, args.warmup, leg=leg_prefix + "_headline")
    # the other witness placement, same K (outside the headline's timed region)
    leg_marker(zk, torch, leg_prefix + "_between")
    other_elapsed, other_stage = timed_run(not headline_hbm, args.steps, 1, leg=leg_prefix + "_other_witness_placement")

    def one_proof(i):
        return prover.prove_dev(wits_dev[i % len(wits_dev)].data_ptr())

    # N > 1: one more proof with fixed (r, s), checked on rank 0 against an UNSHARDED prover of the same
    # key on rank 0's GPU (tables as in the zkey, whole chain on one GPU): the multi-GPU path must
    # reproduce the single-GPU proof bit for bit.  Outside the timed region.
    verified = None
    if world > 1 and args.verify:
        vr, vs = 0x1234567, 0x7654321
        if chain is not None:
            chain.submit(wtns=wits_host[0], r=vr, s=vs)
        else:
            prover.submit_host(wits_host[0], vr, vs)
        parts = zk.gather_partials(prover.collect_msm(), dist, xdev)
        if rank == 0:
            sharded = prover.prove_finish(parts, vr, vs)
            ref = ProverFromView(zk, wl, device=local_rank, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=False)
            verified = ref.prove_host(wits_host[0], vr, vs) == sharded
            ref.lib.zk_prover_destroy(ref.h)

    # N > 1, after the headline: the OTHER way to use N GPUs — every rank proves its own proofs on its own GPU with an
    # unsharded prover, no collective in the data path (what proverServer's throughput mode runs; "replicas
    # only").  Same timing rules: barrier + synchronize on both sides, max over ranks; value = N * K proofs / that time.
    replicas = None
    if world > 1 and not args.no_replicas:
        replicas = replicas_leg(zk, wl, local_rank, k, wits_host, args.steps, args.precomp, dist, xdev, torch, world)

    latency_ms = latency_host_ms = None
    lone = {}
    if world == 1:                        # outside the timed region: strictly one proof at a time
        torch.cuda.synchronize()
        leg_marker(zk, torch, leg_prefix + "_lone_resident")
        t1 = time.perf_counter()
        for i in range(3):
            one_proof(i)
            for kk, v in prover.timings().items():
                lone[kk] = lone.get(kk, 0.0) + v / 3
        latency_ms = (time.perf_counter() - t1) / 3 * 1e3
        leg_marker(zk, torch, leg_prefix + "_lone_host_witness")
        t1 = time.perf_counter()
        nsync = 8                       # (three were too few: the mean moved by +-1 ms between runs of one box)
        for i in range(nsync):
            prover.prove_host(wits_host[i % len(wits_host)])        # zk_prove: host witness, synchronous (main_prover.cpp:75)
        latency_host_ms = (time.perf_counter() - t1) / nsync * 1e3
    leg_marker(zk, torch, leg_prefix + "_after")
    if dist:
        dist.barrier()
    if rank != 0:
        prover.lib.zk_prover_destroy(prover.h)
        if dist:
            dist.destroy_process_group()
        return None

    ms_per_step = elapsed / args.steps * 1e3
    shape_txt = ("domainSize=nVars=2^%d, nPublic=1" % k) if args.shape == "dense" else \
        ("domainSize=2^%d, nVars=%d, nPublic=%d, ~30%% of the rows of A and of B1/B2 at infinity" % (k, wl["nVars"], wl["nPublic"]))
    config = {"workload": "synthetic BN254 zkey, 2^%d constraints (%s, nCoefs=%d), %s witness" % (k, shape_txt, wl["nCoefs"], "uniform random" if args.witness == "uniform" else "realistic (80% {0,1}, 15% <2^32, 5% full)"),
              "shape": args.shape, "log2n": k,
              "parallelism": "msm-point-shard x%d" % world + (", chain partitioned (4 x all_to_all per proof)" if partitioned else (", chain replicated" if world > 1 else "")),
              "window_bits": plan["window_bits_h"], "windows": plan["windows_h"], "precomputed_window_tables": bool(plan["precomputed_tables"]), "table_rows": plan.get("table_rows_h", 0), "bucket_sets_per_msm": plan.get("bucket_sets_h", 0),
              "witness_msm_window_bits": plan["window_bits_w"] if plan["window_bits_w"] != plan["window_bits_h"] else None,
              "msm_a_b1_c_in_one_launch": bool(plan["msm_a_b1_c_one_launch"]), "lanes": plan["lanes"],
              "proofs_in_flight": depth_for(headline_hbm) if pipelined else 1,
              "witnesses_per_submission": args.batch if (args.batch > 1 and world == 1 and not headline_hbm) else 1,
              "host_threads": 2 if (pipelined and world == 1 and args.collector_thread) else 1,
              "witness": "hbm-resident" if headline_hbm else "host-pageable, upload timed",
              "witness_upload": ("each rank uploads 1/N over PCIe, all_gather over xGMI" if sliced_upload else "whole witness per rank") if world > 1 else "whole witness",
              "distinct_witnesses": len(wits_host)}
    # Dominant kernel by instructions and by exclusive time: k_msm_accum_l1<Fq>, the G1 bucket accumulation of MSM A, B1, C, H.
    # launch_ms = its time PER G1 MSM (hipEvents on the kernel's own stream around every launch, summed over a proof, / 4): the
    # unit of the 96*n algorithmic bytes (64 B point + 32 B scalar), whatever the number of launches the four MSMs share.
    g1_ms, g2_ms = stage["g1_l1_kernel"], stage["g2_l1_kernel"]
    pts = n / world
    alg1, alg2 = G1_MSM_BYTES_PER_POINT * pts, G2_MSM_BYTES_PER_POINT * pts
    ck = clock.summary()
    roofline = {"bound": "hbm", "kernel": "k_msm_accum_l1<Fq>", "achieved": round(alg1 / (g1_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg1 / (g1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "traffic_source": None, "traffic_ratio": None,
                "launch_ms": round(g1_ms, 4), "algorithmic_bytes": int(alg1), "msms_per_proof": 4, "launches_per_proof": 2 if plan["msm_a_b1_c_one_launch"] else 4,
                "launch_ms_one_in_flight": round(lone["g1_l1_kernel"], 4) if lone else None,
                "frac_one_in_flight": round(alg1 / (lone["g1_l1_kernel"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if lone else None,
                "g2_kernel": "k_msm_accum_l1_g2s", "g2_launch_ms": round(g2_ms, 4), "g2_launch_ms_one_in_flight": round(lone["g2_l1_kernel"], 4) if lone else None,
                "g2_algorithmic_bytes": int(alg2), "g2_frac": round(alg2 / (g2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "clock_ghz": ck.get("clock_ghz"), "power_w": ck.get("power_w"), "clock_samples": ck.get("samples"),
                "simds": torch.cuda.get_device_properties(local_rank).multi_processor_count * 4,
                # what the kernel's largest launch gathers from: W rows of 64 B per point of every table the launch walks
                "gather_table_bytes": int((3 if plan["msm_a_b1_c_one_launch"] else 1) * max(1, plan.get("table_rows_h", 1)) * pts * 64),
                "whole_proof_algorithmic_bytes": 1424 * n, "whole_proof_frac": round(1424 * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}

    out = {
        "metric": "Groth16 proofs/sec", "value": round(args.steps / elapsed, 4), "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i64 (v_mad_i64_i32 column sums over 9x29-bit signed limbs, BN254 Fr/Fq)",
        "data": "synthetic",
        "config": config,
        "roofline": roofline,
        "stage_ms": {kk: round(v, 3) for kk, v in stage.items()},
        "setup_s": {"generate": round(t_gen, 2), "create": round(t_create, 2)},
    }
    other = {"value": round(args.steps / other_elapsed, 4), "unit": "proofs/s", "ms_per_step": round(other_elapsed / args.steps * 1e3, 3), "steps": args.steps}
    out["host_witness" if headline_hbm else "resident_witness"] = other
    if verified is not None:
        out["multi_gpu_proof_equals_single_gpu_proof"] = verified
    if latency_ms is not None:
        out["latency_ms_one_at_a_time"] = {"witness_in_hbm": round(latency_ms, 3), "witness_in_host_memory": round(latency_host_ms, 3)}
        out["ms_per_proof_sync"] = round(latency_host_ms, 3)       # SURVEY 8(d)'s ms/proof: one synchronous zk_prove, witness in host RAM
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(wl, k, synth, prover, wits_host[0], wits_dev[0], args.cpu_budget_s)
    if world > 1:
        out["first_contact"] = first_contact_info
        out["rccl_ranks"] = world
        out["exchange_backend"] = "gloo via the host (ZK_BENCH_SHARE_GPU test hook)" if share else "nccl (RCCL over xGMI)"
        if chain is not None:
            out["exchange_ms"] = chain.phase_times_ms()
        if replicas is not None:
            out["replicas"] = replicas
    prover.lib.zk_prover_destroy(prover.h)
    if dist:
        dist.destroy_process_group()
    if world == 1 and args.shape == "dense" and args.witness == "uniform" and k >= 20 and want_shard8(args):
        try:
            out["shard8"] = shard8_leg(zk, torch, wl, k, wits_dev[0], ms_per_step, latency_ms, args.precomp)
        except Exception as exc:           # noqa: BLE001  (a failed probe must not cost the line)
            out["shard8"] = {"log2n": k, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    return out


def want_shard8(args):
    return not args.no_shard8 and args.log2n in (22, 24) and args.batch <= 1 and not args.window_bits


def shard8_leg(zk, torch, wl, k, w_dev, single_ms, single_lone_ms, precomp, ranks=8):
    """ONE rank's share of a proof split across `ranks` GPUs, measured on this GPU (what tools/shard_probe.py measures): a prover
    created as shard 0 of `ranks` with the chain partitioned (its eighth of every MSM table — window-precomputed —, of the rows of
    A.w / B.w and of the transforms) runs its phases through zk_shard_* with the all_to_all LEFT OUT: the exchange buffers keep
    whatever they hold, so the result is meaningless, but kernels, sizes and launch counts are exactly a rank's.  Missing from the
    figure: four rounds of all_to_all per proof (2 x 7/8 of a block per transform over xGMI) and the 384-byte gather of the sums."""
    from rapidsnark_old_amd.dist import ShardedChain
    p = ProverFromView(zk, wl, device=0, shard_index=0, shard_count=ranks, window_bits=0, timings=True, precomp=precomp, partitioned_chain=True)
    try:
        plan = p.info()
        ch = ShardedChain(p.lib, p.h, None, torch.device("cuda:0"), exchange=lambda dst, src: None)
        ch.time_phases = False

        def submit():
            ch.submit(d_wtns=w_dev.data_ptr())

        for _ in range(3):
            submit()
            p.collect_msm()
        launches = p.info()["kernel_launches_last_proof"]
        torch.cuda.synchronize()
        leg_marker(zk, torch, "2p%d_shard8_one_at_a_time" % k)      # (tools/profile_bench.sh: a kernel table per mode of this leg)
        n1 = 8
        t0 = time.perf_counter()
        for _ in range(n1):
            submit()
            p.collect_msm()
        one = (time.perf_counter() - t0) / n1 * 1e3
        stage = {a: round(b, 3) for a, b in p.timings().items()}
        leg_marker(zk, torch, "2p%d_shard8_two_in_flight" % k)
        n2 = 12
        submit()
        t0 = time.perf_counter()
        for _ in range(n2):
            submit()
            p.collect_msm()
        p.collect_msm()
        two = (time.perf_counter() - t0) / (n2 + 1) * 1e3
        leg_marker(zk, torch, "2p%d_shard8_after" % k)
        launches_busy = p.info()["kernel_launches_last_proof"]
    finally:
        p.lib.zk_prover_destroy(p.h)
    ideal = single_ms / ranks
    return {"log2n": k, "ranks": ranks, "rank": 0, "chain": "partitioned", "measured_on": "one GPU: this rank's kernels only, the four all_to_all rounds left out",
            "rank_share_ms_one_at_a_time": round(one, 3), "rank_share_ms_two_in_flight": round(two, 3),
            "single_gpu_ms_per_step": round(single_ms, 3), "single_gpu_ms_one_at_a_time": round(single_lone_ms, 3) if single_lone_ms else None,
            "ideal_share_ms": round(ideal, 3),
            "implied_speedup_one_at_a_time": round((single_lone_ms or single_ms) / one, 2), "implied_speedup_two_in_flight": round(single_ms / two, 2),
            "kernel_launches_per_rank": launches, "kernel_launches_per_rank_two_in_flight": launches_busy,
            "window_bits_h": plan["window_bits_h"], "additions_per_point_h": plan["windows_h"],
            "window_bits_w": plan["window_bits_w"], "additions_per_point_w": plan["windows_w"],
            "stage_ms_one_at_a_time": stage}


def finish_roofline(out, cs, source, ceiling):
    """Counter-derived fields of one leg's roofline (cs: rapidsnark_old_amd.counters summary or None): HBM traffic per G1 MSM and its
    ratio to the algorithmic bytes, the random-gather rate of the one-in-flight launch against the chip's ceiling for 64-byte
    bursts (tools/gather_probe, committed), and the path's own bound — VALU issue at 4 cycles per wave-level instruction."""
    r = out["roofline"]
    r["traffic_source"] = source
    cs = cs or {}
    t1 = cs.get("g1_hbm_bytes_per_msm")
    r["traffic"] = t1
    r["traffic_ratio"] = round(t1 / r["algorithmic_bytes"], 3) if t1 else None
    r["g2_traffic"] = cs.get("g2_hbm_bytes_per_launch")
    r["transforms_traffic_per_proof"] = cs.get("transforms_hbm_bytes_per_proof")
    lm = r.get("launch_ms_one_in_flight")
    r["gather_bytes_per_s"] = round(t1 / (lm * 1e-3)) if (t1 and lm) else None
    r["gather_ceiling_bytes_per_s"] = round(ceiling["bytes_per_s"]) if ceiling else None
    r["gather_ceiling_source"] = ("%s: 64-B rows, %d MB table, best access pattern" % (ceiling["source"], ceiling["table_mb"])) if ceiling else None
    r["gather_frac"] = round(r["gather_bytes_per_s"] / ceiling["bytes_per_s"], 4) if (ceiling and r["gather_bytes_per_s"]) else None
    instr = cs.get("valu_instructions_per_proof")
    r["valu_instructions_per_proof"] = instr
    r["cycles_per_instruction"] = 4.0
    r["g1_valu_per_msm"], r["g2_valu_per_launch"], r["transforms_valu_per_proof"] = cs.get("g1_valu_per_msm"), cs.get("g2_valu_per_launch"), cs.get("transforms_valu_per_proof")
    ghz = r.get("clock_ghz")
    r["issue_bound_ms"] = round(instr * 4.0 / (r["simds"] * ghz * 1e9) * 1e3, 3) if (instr and ghz) else None
    r["issue_frac"] = round(r["issue_bound_ms"] / out["ms_per_step"], 4) if r["issue_bound_ms"] else None


def summary_of(out):
    """The scalars DESIGN.md section 6 quotes, flat and LAST in the line."""
    r = out["roofline"]
    s = {"proofs_per_s": out["value"], "ms_per_step": out["ms_per_step"], "ms_per_proof_sync": out.get("ms_per_proof_sync"),
         "ms_per_step_resident": out.get("resident_witness", {}).get("ms_per_step"),
         "roofline_frac": r["frac"], "roofline_frac_one_in_flight": r.get("frac_one_in_flight"), "g1_launch_ms": r["launch_ms"],
         "g1_launch_ms_one_in_flight": r.get("launch_ms_one_in_flight"), "g2_launch_ms": r["g2_launch_ms"], "g2_launch_ms_one_in_flight": r.get("g2_launch_ms_one_in_flight"),
         "traffic_ratio": r.get("traffic_ratio"), "gather_frac": r.get("gather_frac"), "whole_proof_frac": r["whole_proof_frac"],
         "valu_instructions_per_proof": r.get("valu_instructions_per_proof"), "issue_bound_ms": r.get("issue_bound_ms"), "issue_frac": r.get("issue_frac"),
         "clock_ghz": r.get("clock_ghz"), "power_w": r.get("power_w"),
         "counters": "measured in this run" if str(r.get("traffic_source", "")).startswith("measured") else "replayed"}
    for key, tag in (("also_2p20", "2p20"), ("also_realistic", "realistic")):
        o = out.get(key)
        if o:
            s["ms_per_step_" + tag], s["proofs_per_s_" + tag], s["ms_per_proof_sync_" + tag] = o["ms_per_step"], o["value"], o.get("ms_per_proof_sync")
            s["issue_frac_" + tag] = o["roofline"].get("issue_frac")
    c = out.get("cpu_baseline")
    if c:
        s["cpu_proofs_per_s"], s["cpu_cores"], s["cpu_variant"] = c["value"], c["cores"], c.get("variant")
        s["gpu_over_cpu"] = round(out["value"] / c["value"], 1) if c["value"] else None
        s["gpu_proof_bit_exact_vs_cpu"] = c.get("gpu_proof_bit_exact_vs_cpu")
    sv = out.get("also_server")
    if sv:
        s["server_proofs_per_s"] = sv.get("value")
    o = out.get("also_2p24")
    if o and o.get("value"):
        s["ms_per_step_2p24"], s["proofs_per_s_2p24"], s["ms_per_proof_sync_2p24"] = o["ms_per_step"], o["value"], o.get("ms_per_proof_sync")
    for tag, o in (out.get("also_shard8") or {}).items():
        if "rank_share_ms_two_in_flight" in o:
            s["shard8_%s_rank_ms_one_at_a_time" % tag], s["shard8_%s_rank_ms_two_in_flight" % tag] = o["rank_share_ms_one_at_a_time"], o["rank_share_ms_two_in_flight"]
            s["shard8_%s_ideal_ms" % tag], s["shard8_%s_implied_speedup" % tag] = o["ideal_share_ms"], o["implied_speedup_two_in_flight"]
            s["shard8_%s_launches_per_rank" % tag], s["shard8_%s_additions_per_point" % tag] = o["kernel_launches_per_rank"], o["additions_per_point_h"]
    return s


def counters_child(args):
    """The process rocprofv3 --pmc wraps (rapidsnark_old_amd.counters.run_pass): for every leg, a prover of that configuration,
    a begin marker, 1 + `--counters-proofs` proofs with the witness resident, each submitted beside one more in flight, an end marker."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import rapidsnark_old_amd as zk
    from rapidsnark_old_amd import synth, counters
    legs = [x for x in args.counters_child.split(",") if x]
    spec = {v: kk for kk, v in LEG_OF.items()}
    for i, leg in enumerate(legs, 1):
        shape, k = spec[leg]
        realistic = shape == "circuit"
        wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes(), shape=shape)
        p = ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=args.precomp,
                           sparse_witness=realistic and bool(args.precomp))
        w = torch.from_numpy(synth.make_witness(k, seed=1, kind="realistic" if realistic else "uniform", n_vars=wl["nVars"])).cuda()
        torch.cuda.synchronize()
        # the proofs between the markers are submitted while another one is in flight (uncollected), like every proof of the
        # timed legs but the first: the library gives such a proof fewer, longer level-1 lanes (fewer partial sums to merge)
        p.submit_dev(w.data_ptr())
        marker(zk, torch, counters.MARK_BEGIN + i)
        for _ in range(1 + args.counters_proofs):
            p.submit_dev(w.data_ptr())
            p.collect()
        marker(zk, torch, counters.MARK_END + i)
        p.collect()
        p.lib.zk_prover_destroy(p.h)
        del w, wl
    print("[counters-child] done", flush=True)


def marker(zk, torch, workgroups):
    """A recognisable launch: zk_fr_mul_vec over 256 x workgroups elements = a k_mul_vec<Fr> grid of that many workgroups."""
    torch.cuda.synchronize()
    a = np.zeros(32 * 256 * workgroups, dtype=np.uint8)
    zk.fr_mul_vec(a, a)
    torch.cuda.synchronize()


def replicas_leg(zk, wl, device, k, wits_host, steps, precomp, dist, xdev, torch, world):
    """N > 1: K proofs per GPU on N independent unsharded provers (host witnesses, the default number in flight, one host
    thread per rank) -> the line's `replicas` object.  Weak scaling: per-GPU work is fixed as N grows."""
    p = ProverFromView(zk, wl, device=device, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=precomp)
    depth = p.info()["depth_host_witness"]
    for i in range(depth):                      # untimed: every proof slot exists afterwards
        p.submit_host(wits_host[i % len(wits_host)])
    for i in range(depth):
        p.collect()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flying = 0
    for i in range(steps):
        p.submit_host(wits_host[i % len(wits_host)])
        flying += 1
        if flying == depth:
            p.collect()
            flying -= 1
    while flying:
        p.collect()
        flying -= 1
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=xdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    p.lib.zk_prover_destroy(p.h)
    return {"value": round(world * steps / dt, 4), "unit": "proofs/s", "scaling": "weak", "steps_per_gpu": steps, "ms_per_step_per_gpu": round(dt / steps * 1e3, 3),
            "proofs_in_flight_per_gpu": depth,
            "note": "every GPU proves its own proofs on an unsharded prover (no collective in the data path; proverServer's throughput mode); "
                    "the headline above is ONE proof at a time across all GPUs (north_star: MSMs and NTT partitioned)"}


def multi_gpu_first_contact(args, dist, torch, dev, xdev, rank, world, share):
    """N > 1, BEFORE anything is built or timed: rendezvous with a deadline, then rapidsnark_old_amd.dist.first_contact — the
    peer-access matrix, one all_to_all_single of the proof's real exchange size checked byte for byte against a host
    reference, the 384-byte all_gather.  No multi-GPU path of this repository has run on more than one physical GPU
    (DESIGN.md section 7): the first such run must say what it found instead of hanging or printing a wrong number.  Any
    failure (a rank that never arrives, a collective that errors or delivers wrong bytes) ends EVERY rank with a non-zero
    exit code; rank 0 prints the reason as its one JSON line."""
    import datetime
    import threading

    def fail(reason, info=None):
        if rank == 0:
            print(json.dumps({"metric": "Groth16 proofs/sec", "value": None, "unit": "proofs/s", "n_gpus": world, "error": reason,
                              "first_contact": info, "note": "multi-GPU first-contact check failed before the timed region (bench.py, multi_gpu_first_contact)"}), flush=True)
        sys.stderr.write("[bench] rank %d: %s\n" % (rank, reason))
        sys.stderr.flush()
        os._exit(3)

    deadline_s = float(os.environ.get("ZK_BENCH_RENDEZVOUS_S", "240"))
    watchdog = threading.Timer(deadline_s + 60, lambda: fail("no progress for %.0f s in the rendezvous / first collectives (a rank missing or a hung collective)" % (deadline_s + 60)))
    watchdog.daemon = True
    watchdog.start()
    try:
        to = datetime.timedelta(seconds=deadline_s)
        if share:
            dist.init_process_group(backend="gloo", timeout=to)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=to)   # "nccl" is RCCL on ROCm
    except Exception as exc:           # noqa: BLE001
        fail("init_process_group failed: %s" % exc)
    info = None
    try:
        from rapidsnark_old_amd.dist import first_contact
        partitioned = world in (2, 4, 8) and args.chain != "replicated" and args.log2n >= 6
        nloc = (1 << args.log2n) // world
        xbytes = 3 * nloc * 32 if partitioned else 384 * world          # the chain's exchange buffer (a|b|c blocks), else just the records
        xbytes = max(world, xbytes - xbytes % world)
        exchange = None
        if share:                        # single-GPU test hook: gloo on host tensors (all_gather, as the chain's own hook)
            def exchange(dst, src):
                allb = [torch.empty_like(src) for _ in range(world)]
                dist.all_gather(allb, src)
                part = src.numel() // world
                dst.copy_(torch.cat([allb[sidx][rank * part:(rank + 1) * part] for sidx in range(world)]))
        info = first_contact(dist, xdev, rank, world, xbytes, exchange=exchange)
        info["chain_partitioned"] = bool(partitioned)
        if not share:
            nd = torch.cuda.device_count()
            info["peer_access"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(nd)] for i in range(nd)]
            info["visible_devices"] = nd
    except Exception as exc:           # noqa: BLE001
        fail("first contact failed: %s: %s" % (type(exc).__name__, exc), info)
    watchdog.cancel()
    if rank == 0:
        sys.stderr.write("[bench] first contact: %s\n" % json.dumps(info))
    return info


_LEGS = []


def leg_marker(zk, torch, name):
    """ZK_BENCH_LEG_MARKERS=1 (tools/profile_bench.sh): a recognisable launch between the legs of a run — zk_fr_mul_vec over
    256 x (16 + leg) elements = a k_mul_vec<Fr> grid of 16 + leg workgroups, which the kernel trace records — so that a
    rocprofv3 --kernel-trace of the DRIVER'S OWN COMMAND can be cut into one kernel table per leg (tools/leg_stats.py).
    Off (no launch, no synchronisation) in a normal run."""
    if os.environ.get("ZK_BENCH_LEG_MARKERS") != "1":
        return
    torch.cuda.synchronize()
    _LEGS.append(name)
    marker(zk, torch, 16 + len(_LEGS))
    sys.stderr.write("[bench] leg marker %d (grid of %d workgroups): %s\n" % (len(_LEGS), 16 + len(_LEGS), name))


from rapidsnark_old_amd.views import view_from_workload, ProverFromView, MultiProverFromView      # noqa: E402,F401  (the wrappers live in the package)


def effective_cores():
    """CPUs this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(wl, k, synth, prover, w0, w0_dev, budget_s):
    """The ONLY place bench.py touches oracle/: the C restatement of rapidsnark's CPU algorithm (oracle/c/zk_oracle.c; OpenMP at
    the reference's own parallel-for sites) timed on this box's host cores on the SAME workload and witness as the GPU, and used
    as the bit-exact checker of the GPU proof for the same (witness, r, s).  Two builds of its field product are timed: `adx`
    (mulx + adcx/adox carry chains: the instruction mix of the reference's ffiasm assembly, README.md:67-69) — the `value` when
    the CPU has BMI2 + ADX — and `generic` (portable C over unsigned __int128).  One small warm-up proof, then as many full
    proofs as the budget holds (median); when a full proof does not fit, the largest member of the family that does, scaled."""
    cores = effective_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)          # before libgomp is loaded: no oversubscription
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    from oracle import c_oracle as co
    odir = os.path.join(ROOT, "oracle", "_build")
    libs = []                                            # (variant, library), fastest first
    for variant, adx in (("adx", True), ("generic", False)):
        if adx and not co.cpu_has_adx():
            continue
        try:
            name = "libzkoracle_native%s.so" % ("_adx" if adx else "")
            co.build(march="native", out="_build/" + name, adx=adx)
            libs.append((variant, co.load(os.path.join(odir, name))))
        except Exception:           # noqa: BLE001  (no compiler on the box: the prebuilt portable library)
            continue
    if not libs:
        libs = [("generic", co.load())]
    r, s = 0x1234567, 0x7654321
    kp = min(k, 16)
    res = {}
    proof_gpu = prover.prove_host(w0, r, s)              # the GPU proof through the reference's own entry point (host witness)
    left = budget_s
    for vi, (variant, lib) in enumerate(libs):
        co._LIB = lib
        co.set_num_threads(cores)
        cores = co.num_threads()                         # threads actually used (= the CPU quota of this box)
        wlp = co.synth_workload(kp)
        wp = synth.make_witness(kp)
        co.prove(co.ZkeyView(wlp), wp, 1, 2)             # warm-up: threads, page tables, caches
        t = time.perf_counter()
        co.prove(co.ZkeyView(wlp), wp, 1, 2)
        t_probe = time.perf_counter() - t
        est = t_probe * (1 << (k - kp)) * 0.8            # (a 2^16 proof is less efficient per constraint than a 2^22 one)
        # the fastest variant gets up to 60 % of the budget (as many full proofs as fit, at most three); the other one what is left
        share = left * 0.6 if (vi == 0 and len(libs) > 1) else left
        if est <= share:
            view, times, proof_cpu = co.ZkeyView(wl), [], None
            while len(times) < 3 and (not times or sum(times) + max(times) <= share):
                t = time.perf_counter()
                proof_cpu = co.prove(view, w0, r, s)
                times.append(time.perf_counter() - t)
            left -= sum(times)
            dt = sorted(times)[len(times) // 2]
            res[variant] = {"s_per_proof": round(dt, 3), "proofs": len(times), "bit_exact": proof_cpu == proof_gpu,
                            "sample": "%d full proof(s) of the same 2^%d workload and witness after a 2^%d warm-up; median %.2f s" % (len(times), k, kp, dt)}
        elif vi == 0:                                    # too slow for the budget: the largest size that fits, scaled linearly in n
            ks = kp
            while ks < k and t_probe * (1 << (ks + 1 - kp)) * 1.15 <= share / 3:
                ks += 1
            wls = co.synth_workload(ks) if ks != kp else wlp
            ws = synth.make_witness(ks)
            times = []
            for _ in range(3):
                t = time.perf_counter()
                co.prove(co.ZkeyView(wls), ws, r, s)
                times.append(time.perf_counter() - t)
            left -= sum(times)
            dt = sorted(times)[1] * (1 << (k - ks))
            res[variant] = {"s_per_proof": round(dt, 3), "proofs": 3, "bit_exact": None,
                            "sample": "median of 3 proofs of the 2^%d member of the same family, scaled x%d to 2^%d (linear in n)" % (ks, 1 << (k - ks), k)}
    best = libs[0][0]
    out = {"value": round(1.0 / res[best]["s_per_proof"], 5), "unit": "proofs/s", "cores": cores, "kind": "port", "variant": best,
           "sample": res[best]["sample"], "s_per_proof": res[best]["s_per_proof"], "gpu_proof_bit_exact_vs_cpu": res[best]["bit_exact"]}
    if "generic" in res and best != "generic":
        out["generic_s_per_proof"], out["generic_bit_exact"] = res["generic"]["s_per_proof"], res["generic"]["bit_exact"]
    out["note"] = "oracle/c/zk_oracle.c, gcc -O3 -march=native -fopenmp; parity unpinned at the reference boundary (DESIGN.md section 2)"
    return out


def server_leg():
    """BASELINE configs[4] on one GPU, proxy key: tools/server_bench.py drives the proverServer executable over REST (keep-alive
    clients, /witness) on a Semaphore-class zkgen key and checks every proof; -> the also_server object, or why it did not run."""
    import subprocess
    exe = os.path.join(ROOT, "rapidsnark-old_amd", "proverServer")
    tool = os.path.join(ROOT, "tools", "server_bench.py")
    if not (os.path.exists(exe) and os.path.exists(tool)):
        return {"value": None, "error": "proverServer / tools/server_bench.py not built"}
    # (without the OpenMP variables the CPU leg exported for the oracle: OMP_PROC_BIND pins a process that loads libgomp —
    # the load generator's sixteen client threads then share one core and the "server" rate halves)
    env = {kk: v for kk, v in os.environ.items() if not kk.startswith("OMP_")}
    try:
        r = subprocess.run([sys.executable, tool, "15", "512", "0", "witness", "semaphore"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           timeout=150, text=True, cwd=ROOT, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"value": None, "error": "server_bench rc %d" % r.returncode}
        d = json.loads(line[-1])
        return {"value": d["proofs_per_s"], "unit": "proofs/s", "n_gpus": 1, "log2n": d["log2n"], "requests": d["requests"], "route": d["route"],
                "succeeded": d["succeeded"], "proofs_equal_to_the_trapdoor_prediction": d["proofs_equal_to_the_trapdoor_prediction"],
                "key": d.get("key", "zkgen proxy key (no Semaphore / iden3-auth zkey exists in the image)"), "ms_per_proof": d["ms_per_proof"]}
    except Exception as exc:           # noqa: BLE001
        return {"value": None, "error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}


if __name__ == "__main__":
    main()
