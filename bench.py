#!/usr/bin/env python
"""bench.py — Groth16 prove() throughput on MI355X (BASELINE.json metric).

A "step" is one whole proof (src/groth16.cpp:48-254) of a synthetic BN254 data set with
n = domainSize = nVars = 2^k (default k = 22: BASELINE configs[2], the largest single-GPU
configuration and the one the 10x target is quoted on; --log2n 20 gives configs[1]).
Every step has its own witness (distinct per step while they fit 8 GiB of host memory: 33 x 128 MiB at
2^22; `config.distinct_witnesses` says how many were made) in HOST memory (pageable numpy arrays), as the
reference's Prover::prove(FrElement *wtns) contract has it (src/groth16.hpp:101,
src/main_prover.cpp:74-75): the upload of every witness is INSIDE the timed region
(zk_prove_submit stages it through pinned memory on a stream of its own, so that it overlaps
the previous proof).  The rate with witnesses already resident in HBM is measured too and
reported as the extra key `resident_witness` (--witness-in hbm makes it the headline again).

  python bench.py --gpus 1 --steps K --warmup W                      (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

N > 1: one process per GPU; every MSM point table is sharded by index range and the
chain (A.w/B.w rows + the six transforms) is PARTITIONED the same way when N is 2, 4 or 8: rank g
holds block g of a, b, c, h, runs the local stages itself and meets the others in the log2(N) top
stages through four rounds of RCCL all_to_all per proof (2 x (N-1)/N of a block per transform over
xGMI; rapidsnark_old_amd.dist.ShardedChain).  The MSM results are exchanged by one all_gather of
the 384-byte partial-sum record; rank 0 adds the partials and does the O(1) final assembly.
--chain replicated keeps the round-1 behaviour (every rank runs the whole chain, no all_to_all).
One proof is split across the ranks => "scaling": "strong".

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel BY TOTAL TIME: the G1 bucket
accumulation k_msm_accum_l1<Fq>, four launches per proof, algorithmic bytes 96*n per launch,
SURVEY §8d; the longest single launch, the G2 accumulation, is reported under `also`) and, at
N = 1, `cpu_baseline` (the C restatement of rapidsnark's CPU algorithm, oracle/, timed on the
host cores: one warm-up, then the median of as many full proofs as the budget holds).
BASELINE's metric is quoted "at 2^20 and 2^22": the default N = 1 run therefore also times the 2^20 member of the
family (configs[1]) the same way and reports it under `also_2p20` (own ms_per_step, synchronous ms/proof and G1-launch
roofline fraction; --no-2p20 skips it, and so do --no-cpu and the other non-default workloads: A/B and profiling runs
want one size).  `ms_per_proof_sync` is SURVEY section 8(d)'s definition of ms/proof: ONE
synchronous zk_prove with the witness in host memory (the reference's main_prover.cpp:75), next to the pipelined period
`ms_per_step`.  N > 1 prints `rccl_ranks` and the mean GPU time of each all_to_all / all_gather phase (`exchange_ms`).
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
                         "the reference's makeProver it is outside the timed prove()); 0 = tables exactly as in the zkey")
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
    ap.add_argument("--traffic-bytes", type=float, default=None, help="HBM bytes per dominant-kernel launch from a PMC run")
    return ap.parse_args()


def main():
    args = parse()
    out = run(args)
    if out is None:                       # ranks > 0
        return
    default_run = args.gpus == 1 and args.log2n == 22 and not args.no_cpu and args.batch <= 1 and args.witness == "uniform" and args.shape == "dense"
    if default_run and not args.no_realistic:
        # what a REAL circom key and witness look like to the prover (BASELINE configs[4]'s fidelity; SURVEY section 8d's
        # secondary line): the circuit-shaped member of the family with the 80/15/5 witness, timed the same way by the same code
        import copy
        a3 = copy.copy(args)
        a3.witness, a3.shape, a3.no_cpu, a3.no_2p20, a3.no_realistic, a3.in_flight = "realistic", "circuit", True, True, True, 0
        o3 = run(a3)
        out["also_realistic"] = {kk: o3[kk] for kk in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_per_proof_sync", "config", "resident_witness",
                                                        "latency_ms_one_at_a_time", "stage_ms") if kk in o3}
        out["also_realistic"]["note"] = ("same 2^22 domain; nVars = 3/4 n + 5, 3 public signals, ~30 % of the rows of A and of B1/B2 at infinity, witness 80 % in "
                                         "{0,1} / 15 % < 2^32 / 5 % full-size: the MSM H and the six transforms cost what they cost in the headline, the four "
                                         "witness MSMs shrink to their non-zero digits")
    if default_run and not args.no_2p20:
        # BASELINE's metric is quoted "at 2^20 and 2^22": configs[1], timed the same way by the same code
        import copy
        a2 = copy.copy(args)
        a2.log2n, a2.no_cpu, a2.no_2p20, a2.in_flight = 20, True, True, 0
        o2 = run(a2)
        out["also_2p20"] = {kk: o2[kk] for kk in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_per_proof_sync", "config", "resident_witness",
                                                   "latency_ms_one_at_a_time", "stage_ms") if kk in o2}
        r2 = o2["roofline"]
        out["also_2p20"]["roofline"] = {kk: r2[kk] for kk in ("kernel", "achieved", "peak", "unit", "frac", "launch_ms", "algorithmic_bytes",
                                                              "launch_ms_one_proof_in_flight", "frac_one_proof_in_flight", "issue_bound", "whole_proof")}
    print(json.dumps(out), flush=True)


def run(args):
    if args.batch > 1 and (args.steps % args.batch or args.gpus != 1):
        raise SystemExit("--batch B needs --gpus 1 and --steps a multiple of B")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # six streams per prover (csrc/prover.hip); read when HIP initialises
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
                            window_bits=args.window_bits, timings=True, precomp=bool(args.precomp), partitioned_chain=partitioned,
                            batch=args.batch if world == 1 else 0, sparse_witness=sparse)
    t_create = time.time() - t0
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
            depth0 = args.in_flight or default_depth(k, in_hbm, world)
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
            depth = args.in_flight or default_depth(k, in_hbm, world)
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
        replicas = replicas_leg(zk, wl, local_rank, k, wits_host, args.steps, bool(args.precomp), dist, xdev, torch, world)

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
    # dominant kernel BY TOTAL TIME: k_msm_accum_l1<Fq> — the G1 bucket accumulation, four launches per
    # proof (MSM A, B1, C, H; 47 % of a proof's VALU instructions).  Duration: hipEvents recorded by the
    # library on the kernel's own stream immediately before/after the launch of MSM A (the other
    # streams' kernels share the chip meanwhile).  Algorithmic bytes: 96 per point (64 B affine point
    # + 32 B scalar, SURVEY §8d "one G1 MSM = 96*n").  `also`: the longest single launch, the G2
    # accumulation of MSM B2 (160 B per point).
    shape_txt = ("domainSize=nVars=2^%d, nPublic=1" % k) if args.shape == "dense" else \
        ("domainSize=2^%d, nVars=%d, nPublic=%d, ~30%% of the rows of A and of B1/B2 at infinity" % (k, wl["nVars"], wl["nPublic"]))
    config = {"workload": "synthetic BN254 zkey, 2^%d constraints (%s, nCoefs=%d), %s witness" % (k, shape_txt, wl["nCoefs"], "uniform random" if args.witness == "uniform" else "realistic (80% {0,1}, 15% <2^32, 5% full)"),
              "shape": args.shape,
              "log2n": k, "parallelism": "msm-point-shard x%d" % world + (", chain partitioned (4 x all_to_all per proof)" if partitioned else (", chain replicated" if world > 1 else "")), "window_bits": args.window_bits or plan_window_bits(n, world, bool(args.precomp)),
              "precomputed_window_tables": bool(args.precomp), "witness_msm_window_bits": 16 if (sparse and k > 18) else None, "proofs_in_flight": (args.in_flight or default_depth(k, headline_hbm, world)) if pipelined else 1,
              "witnesses_per_submission": args.batch if (args.batch > 1 and world == 1 and not headline_hbm) else 1,
              "host_threads": 2 if (pipelined and world == 1 and args.collector_thread) else 1,
              "witness": "resident in HBM before the timed region" if headline_hbm else "pageable host memory; upload inside the timed region (zk_prove_submit)",
              "witness_upload": ("each rank uploads 1/N over PCIe, all_gather over xGMI" if sliced_upload else "whole witness per rank") if world > 1 else "whole witness",
              "distinct_witnesses": "%d for %d steps + %d warm-up%s" % (len(wits_host), args.steps, args.warmup, "" if len(wits_host) >= nw else " (cycled)")}
    batched_abc = batch_abc_default(-(-wl["nVars"] // world), world)
    config["msm_a_b1_c_in_one_launch"] = batched_abc
    g1_ms, g2_ms = stage["g1_l1_kernel"], stage["g2_l1_kernel"]
    pts_per_launch = n / world
    alg_bytes = G1_MSM_BYTES_PER_POINT * pts_per_launch
    achieved = alg_bytes / (g1_ms * 1e-3) / 1e9
    traffic, traffic_src = traffic_from_profiles(args, config, world, "g1")
    traffic2, _ = traffic_from_profiles(args, config, world, "g2")
    issue = issue_bound_from_profiles(config, world, torch.cuda.get_device_properties(local_rank).multi_processor_count, clock.summary(), ms_per_step)
    roofline = {"bound": "hbm", "kernel": "k_msm_accum_l1<Fq> (G1 bucket accumulation; 4 launches per proof: MSM A, B1, C, H — the dominant kernel by VALU "
                                          "instructions (46 % of a proof's) and by exclusive time (4 launches of ~3.8 ms alone out of a ~33 ms period))",
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic, "traffic_source": traffic_src,
                "launch_ms": round(g1_ms, 4), "msms_per_proof": 4, "launches_per_proof": 2 if batched_abc else 4, "algorithmic_bytes": alg_bytes,
                "launch_ms_definition": "time of the kernel per G1 MSM (algorithmic_bytes each): the sum of the kernel's launch durations of a proof / 4"
                                        + (" — MSM A, B1 and C are ONE launch over three point tables (blockIdx.y; 3 x algorithmic_bytes), MSM H another: in a "
                                           "rocprofv3 kernel table of the leg, launch_ms = TotalDurationNs of k_msm_accum_l1<Fq> / (4 x proofs of the leg)" if batched_abc else ""),
                "launch_sharing": "launch_ms is the mean over the timed region, where a launch shares the chip with the kernels of the other proofs in flight "
                                  "(config.proofs_in_flight): more in flight raises proofs/s and LOWERS this fraction; launch_ms_one_proof_in_flight is the same "
                                  "launch with one proof at a time (it still runs beside that proof's own G2 launch), measured after the timed region",
                "launch_ms_one_proof_in_flight": round(lone["g1_l1_kernel"], 4) if lone else None,
                "frac_one_proof_in_flight": round(alg_bytes / (lone["g1_l1_kernel"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if lone else None,
                "also": {"kernel": "k_msm_accum_l1_g2s (G2 bucket accumulation of MSM B2, Fq2 split across lane pairs; the longest single launch)",
                         "launch_ms": round(g2_ms, 4), "algorithmic_bytes": G2_MSM_BYTES_PER_POINT * pts_per_launch,
                         "achieved": round(G2_MSM_BYTES_PER_POINT * pts_per_launch / (g2_ms * 1e-3) / 1e9, 3),
                         "frac": round(G2_MSM_BYTES_PER_POINT * pts_per_launch / (g2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic2},
                "issue_bound": issue,
                "whole_proof": {"algorithmic_bytes": 1424 * n, "achieved": round(1424 * n / (ms_per_step * 1e-3) / 1e9, 2),
                                "frac": round(1424 * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                "note": "B_alg = 1424*n bytes per proof (SURVEY §8d) over the measured period"}}

    out = {
        "metric": "Groth16 proofs/sec", "value": round(args.steps / elapsed, 4), "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32/i64 (BN254 Fr/Fq in 9x29-bit signed limbs, 64-bit v_mad_i64_i32 column accumulators)",
        "data": "synthetic",
        "config": config,
        "roofline": roofline,
        "stage_ms": {kk: round(v, 3) for kk, v in stage.items()},
        "setup_s": {"generate": round(t_gen, 2), "create": round(t_create, 2)},
    }
    other = {"value": round(args.steps / other_elapsed, 4), "unit": "proofs/s", "ms_per_step": round(other_elapsed / args.steps * 1e3, 3),
             "steps": args.steps, "note": "same K proofs, timed the same way, outside the headline's timed region"}
    out["host_witness" if headline_hbm else "resident_witness"] = other
    if verified is not None:
        out["multi_gpu_proof_equals_single_gpu_proof"] = verified
    if latency_ms is not None:
        out["latency_ms_one_at_a_time"] = {"witness_in_hbm": round(latency_ms, 3), "witness_in_host_memory": round(latency_host_ms, 3)}
        # SURVEY section 8(d)'s ms/proof: one synchronous zk_prove, witness in host RAM, nothing else on the GPU
        out["ms_per_proof_sync"] = round(latency_host_ms, 3)
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
    return out


def replicas_leg(zk, wl, device, k, wits_host, steps, precomp, dist, xdev, torch, world):
    """N > 1: K proofs per GPU on N independent unsharded provers (host witnesses, the default number in flight, one host
    thread per rank) -> the line's `replicas` object.  Weak scaling: per-GPU work is fixed as N grows."""
    p = ProverFromView(zk, wl, device=device, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=precomp)
    depth = default_depth(k, False, 1)
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


def default_depth(k, in_hbm, world=1):
    """Proofs in flight per GPU: large circuits saturate the chip with two (three when the witness upload has to be
    hidden); below 2^19 a proof is bound by the serial latency of its ~80 small kernels and more in flight fills the GPU."""
    if k < 19:
        return 8
    if k <= 22 and world == 1:          # four lanes of streams per prover up to 2^22 (csrc/prover.hip): 2^20 11.1 -> 10.2 ms with six in
        return 6                            # flight; 2^22: four / five / six in flight 33.5 / 32.6 / 32.5 ms with host witnesses (r03 A/B, three
                                            # alternations), resident unchanged at 32.5: the deeper pipeline hides the 128 MiB upload completely
    return 2 if in_hbm else 3


def profile_order(path):
    """profiles/<tag>_...: tags run r04a ... r04z, r04aa ... (a longer tag is a later one)"""
    tag = os.path.basename(path).split("_")[0]
    return (len(tag), tag)


def traffic_from_profiles(args, config, world, which):
    """HBM bytes per launch of the roofline kernel.  NOT measured by this run: PMC counters need
    rocprofv3 around the process, so the figure is REPLAYED from the committed counter passes
    (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of this same command;
    profiles/*_pmc_traffic.json) when they were taken on this exact configuration (same workload,
    window bits, table mode, GPU count); else null.  Returns (bytes | None, source string)."""
    if args.traffic_bytes is not None and which == "g1":
        return args.traffic_bytes, "--traffic-bytes (command line)"
    import glob
    keys = ("log2n", "parallelism", "window_bits", "precomputed_window_tables")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), key=profile_order, reverse=True):
        try:
            d = json.load(open(path))
            bc = d.get("bench", {}).get("config", {})
            if all(bc.get(kk) == config.get(kk) for kk in keys) and bc.get("shape", "dense") == config.get("shape") and d.get("bench", {}).get("n_gpus") == world:
                for name, v in d["kernels"].items():
                    is_g2 = "k_msm_accum_l1_g2s" in name or ("k_msm_accum_l1" in name and "Fp2T" in name)
                    is_g1 = "k_msm_accum_l1" in name and not is_g2
                    if (which == "g2" and is_g2) or (which == "g1" and is_g1):
                        raw, how = v["hbm_bytes_raw"], "per launch"
                        if is_g1 and bc.get("msm_a_b1_c_in_one_launch"):
                            # that run launched the kernel twice per proof — once over the three tables of MSM A, B1, C, once for
                            # MSM H — so the mean per LAUNCH is two MSMs' worth: per MSM (the unit of algorithmic_bytes) = x 2 / 4
                            raw, how = int(raw * 2 / 4), "per G1 MSM (mean per launch x 2 launches / 4 MSMs: A, B1, C share one launch)"
                        return raw, "replayed from %s (separate rocprofv3 --pmc passes; raw FETCH_SIZE + WRITE_SIZE %s)" % (os.path.relpath(path, ROOT), how)
        except (OSError, ValueError, KeyError):
            continue
    return None, "no counter pass committed for this configuration"


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


def issue_bound_from_profiles(config, world, cus, clock, ms_per_step):
    """The path's OWN roofline: it is bound by VALU issue, not by HBM.  One wave-level VALU instruction
    of this path occupies its SIMD for ~4 cycles (v_mad_i64_i32 / 64-bit integer rate, profiles/r01_ubench_valu.txt), so
        bound_ms = instructions per proof x 4 cycles / (SIMDs x shader clock).
    Instructions per proof are NOT measured by this run (SQ_INSTS_VALU needs rocprofv3 --pmc around the process): like
    `traffic` they are REPLAYED from the committed counter pass of this exact configuration
    (profiles/*_valu_instruction_budget.json, written by tools/profile_bench.sh); the clock is sampled live while the
    headline's timed region runs (rapidsnark_old_amd.telemetry)."""
    import glob
    keys = ("log2n", "parallelism", "window_bits", "precomputed_window_tables")
    instr = src = per_kernel = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_instruction_budget.json")), key=profile_order, reverse=True):
        try:
            d = json.load(open(path))
            bc = d.get("bench_config", {})
            if all(bc.get(kk) == config.get(kk) for kk in keys) and bc.get("shape", "dense") == config.get("shape") and d.get("n_gpus") == world:
                instr, per_kernel = d["valu_instructions_per_proof"], d.get("kernels")
                src = "replayed from %s (rocprofv3 --pmc SQ_INSTS_VALU pass of this command)" % os.path.relpath(path, ROOT)
                break
        except (OSError, ValueError, KeyError):
            continue
    simds = cus * 4
    ghz = clock.get("clock_ghz")
    out = {"valu_instructions_per_proof": instr, "instructions_source": src or "no SQ_INSTS_VALU pass committed for this configuration",
           "cycles_per_instruction": 4.0, "simds": simds, "clock_ghz": ghz, "power_w": clock.get("power_w"), "clock_source": clock.get("source"),
           "clock_samples": clock.get("samples"), "bound_ms": None, "achieved_frac": None,
           "note": "bound_ms = valu_instructions_per_proof x cycles_per_instruction / (simds x clock_ghz); achieved_frac = bound_ms / ms_per_step "
                   "(1.0 = every SIMD issues a VALU instruction of this path every 4 cycles for the whole period)"}
    if instr and ghz:
        out["bound_ms"] = round(instr * 4.0 / (simds * ghz * 1e9) * 1e3, 3)
        out["achieved_frac"] = round(out["bound_ms"] / ms_per_step, 4)
    if per_kernel:
        out["largest_kernels"] = per_kernel
    return out


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
    n = 256 * (16 + len(_LEGS))
    a = np.zeros(32 * n, dtype=np.uint8)
    zk.fr_mul_vec(a, a)
    torch.cuda.synchronize()
    sys.stderr.write("[bench] leg marker %d (grid of %d workgroups): %s\n" % (len(_LEGS), 16 + len(_LEGS), name))


def batch_abc_default(witness_entries_per_prover, world):
    """Mirror of the library's rule (csrc/prover.hip, zk_prover_create): MSM A, B1 and C as one set of launches unless the
    prover is unsharded and holds 2^20 .. 2^22 - 1 witness entries; ZKHIP_BATCH_ABC overrides."""
    e = os.environ.get("ZKHIP_BATCH_ABC")
    if e is not None:
        return e != "0"
    return not (world == 1 and (1 << 20) <= witness_entries_per_prover < (1 << 22))


def plan_window_bits(n, world, precomp):
    """Mirror of make_msm_plan (csrc/msm.hip)."""
    per = (n + world - 1) // world
    lg = per.bit_length() - 1
    if precomp:
        c = 19 if lg == 20 else max(2, lg - 2)
        return min(20, c)
    return max(2, min(16, lg - 6))


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
    """The ONLY place bench.py touches oracle/: the C restatement of rapidsnark's CPU algorithm
    timed on this box's host cores, on a bounded sample of the same workload, and used as the
    bit-exact checker of the GPU proof for the same (witness, r, s).  Timing: one warm-up proof
    (a smaller member of the family: threads, page tables and caches are up afterwards), then as
    many full proofs as the budget holds (at least one, at most five), median reported."""
    cores = effective_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)          # before libgomp is loaded: no oversubscription
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    from oracle import c_oracle as co
    try:
        co.build(march="native", out="_build/libzkoracle_native.so")
        lib_path = os.path.join(ROOT, "oracle", "_build", "libzkoracle_native.so")
        co.load(lib_path)
        co._LIB = co.load(lib_path)
    except Exception:
        co.load()
    co.set_num_threads(cores)
    cores = co.num_threads()                             # threads actually used (= the CPU quota of this box)
    # warm-up + probe at 2^16 to size the sample
    kp = min(k, 16)
    wlp = co.synth_workload(kp)
    wp = synth.make_witness(kp)
    co.prove(co.ZkeyView(wlp), wp, 1, 2)
    t = time.perf_counter()
    co.prove(co.ZkeyView(wlp), wp, 1, 2)
    t_probe = time.perf_counter() - t
    est_full = t_probe * (1 << (k - kp)) * 1.15
    r, s = 0x1234567, 0x7654321
    note = ("C restatement of rapidsnark's CPU algorithm (oracle/c/zk_oracle.c), gcc -O3 -march=native -fopenmp; "
            "NOT ffiasm: hand-written ADX assembly may be 1.3-2x faster.  Parity unpinned at the reference boundary: the "
            "reference ships no vectors and cannot be built in this image (DESIGN.md section 2), so 'bit-exact' below means "
            "against this restatement, itself pinned by the trapdoor check and third-party vectors only")
    if est_full <= budget_s * 1.5:
        view = co.ZkeyView(wl)
        times, proof_cpu = [], None
        while len(times) < 5 and (not times or sum(times) + max(times) <= budget_s):      # as many as the budget holds
            t = time.perf_counter()
            proof_cpu = co.prove(view, w0, r, s)
            times.append(time.perf_counter() - t)
        runs = len(times)
        dt = sorted(times)[len(times) // 2]
        proof_gpu = prover.prove_host(w0, r, s)                  # the GPU proof through the reference's own entry point (host witness)
        return {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": cores, "kind": "port",
                "sample": "%d full proof(s) of the same 2^%d workload and witness after a 2^%d warm-up; median %.2f s (all: %s)" % (runs, k, kp, dt, ", ".join("%.2f" % x for x in times)),
                "note": note, "gpu_proof_bit_exact_vs_cpu": proof_cpu == proof_gpu}
    # too slow for the budget: largest size that fits, scaled linearly in n
    ks = kp
    while ks < k and t_probe * (1 << (ks + 1 - kp)) * 1.15 <= budget_s / 3:
        ks += 1
    wls = co.synth_workload(ks) if ks != kp else wlp
    ws = synth.make_witness(ks)
    times = []
    for _ in range(3):
        t = time.perf_counter()
        co.prove(co.ZkeyView(wls), ws, r, s)
        times.append(time.perf_counter() - t)
    dt = sorted(times)[1]
    return {"value": round(1.0 / (dt * (1 << (k - ks))), 5), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "median of 3 proofs of the 2^%d member of the same synthetic family (%.2f s), scaled x%d to 2^%d (linear in n)" % (ks, dt, 1 << (k - ks), k),
            "note": note}


if __name__ == "__main__":
    main()
