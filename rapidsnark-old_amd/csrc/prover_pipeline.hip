// One proof as enqueued device work (reference Prover::prove, src/groth16.cpp:48-254):
//   SpMV -> c=a.b -> 3x(DIF iNTT, coset*1/n, DIT NTT) -> h -> digits/sort(w), digits/sort(h)
//   -> bucket accumulation A,B1,C,H (G1) and B2 (G2) -> bucket reduce -> host Horner + assembly,
// split into phases (a prover holding one block of a partitioned chain stops where blocks are exchanged), submit / collect
// with up to ZK_MAX_IN_FLIGHT proofs in flight, and the synchronous entry points over them.
#include "prover_internal.hpp"

// Witness of a host-witness proof -> the slot's HBM copy, on the upload stream.  A caller buffer in
// pinned memory (zk_host_alloc, or registered by the caller) is read by the DMA engine directly; a
// pageable one (the reference's contract: Prover::prove(FrElement *wtns), src/groth16.hpp:101) is
// first copied to the slot's pinned staging buffer BY A HOST FUNCTION ON THE UPLOAD STREAM, so the
// calling thread returns at once and goes on enqueueing the proof (measured at 2^22: staging inside
// the call cost 5 ms per proof — the two-in-flight overlap has to wait for the next submit).  Either
// way the caller's buffer must stay valid and untouched until the proof has been collected.
// The copies are done by a small persistent pool (four threads, started with the first staged witness): a std::thread per
// segment cost ~40 us each, which is what a 4 MiB piece of a 2^20 witness takes to copy — the pieces of such a witness were
// staged by ONE thread each and the upload of a lone 2^20 proof was bound by that memcpy (1.4 ms for 0.65 ms of DMA).
namespace {
struct StagePool {
    struct Seg { uint8_t *dst; const uint8_t *src; size_t len; std::atomic<int> *left; };
    std::mutex m;
    std::condition_variable cv, done;
    std::deque<Seg> q;
    std::vector<std::thread> th;
    bool stop = false;
    StagePool() {
        for (int i = 0; i < 4; i++) th.emplace_back([this] { run(); });
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void run() {
        for (;;) {
            Seg s;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                s = q.front();
                q.pop_front();
            }
            memcpy(s.dst, s.src, s.len);
            if (s.left->fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(m);
                done.notify_all();
            }
        }
    }
    // copies [src, src + bytes) to dst with the pool's threads AND the calling one; returns when all of it is there
    void copy(uint8_t *dst, const uint8_t *src, size_t bytes) {
        const size_t nseg = bytes >= ((size_t)1 << 20) ? 5 : 1, per = (bytes / nseg + 63) & ~(size_t)63;
        std::atomic<int> left{0};
        size_t mine_lo = 0, mine_hi = bytes;
        if (nseg > 1) {
            int pushed = 0;
            std::lock_guard<std::mutex> lk(m);
            for (size_t t = 1; t < nseg; t++) {
                const size_t lo = t * per, hi = lo + per < bytes ? lo + per : bytes;
                if (lo < hi) {
                    q.push_back(Seg{dst + lo, src + lo, hi - lo, &left});
                    pushed++;
                }
            }
            left.store(pushed);
            mine_hi = per < bytes ? per : bytes;
        }
        if (nseg > 1) cv.notify_all();
        memcpy(dst + mine_lo, src + mine_lo, mine_hi - mine_lo);
        if (nseg > 1) {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return left.load() == 0; });
        }
    }
};
StagePool &stage_pool() {
    static StagePool *pool = new StagePool();      // (never destroyed: its threads must not be joined from an exit handler)
    return *pool;
}
}   // namespace
void zkp::stage_job_run(void *arg) {
    const StageJob *j = (const StageJob *)arg;
    stage_pool().copy(j->dst, j->src, j->bytes);
}
// `count` host witnesses (count <= the prover's batch) -> consecutive nVars-element vectors of the slot's buffer; the
// vectors of a batch that are not used are zeroed (an all-zero witness has no non-zero digit: it costs nothing in
// the MSMs).  d_src (batch provers fed a device pointer): vector 0 is copied from device memory instead.
static const Fr *upload_witnesses(zk_prover *p, zk_prover::ProofSlot &q, const uint8_t *const *h_wtns, uint32_t count,
                                  hipEvent_t src_ready = nullptr, const Fr *d_src = nullptr) {
    const size_t bytes = (size_t)p->nVars * 32;
    ensure_witness_buffer(p, q);
    hipStream_t sh = p->stream_h2d;
    if (src_ready) HIP_TRY(hipStreamWaitEvent(sh, src_ready, 0));      // the (pinned) source is still being filled
    const bool tm = (p->flags & ZK_FLAG_TIMINGS) != 0;
    for (uint32_t k = 0; k < count; k++) {
        uint8_t *dst = (uint8_t *)q.wtns_dev.p + (size_t)k * bytes;
        if (d_src) {
            if (tm && k == 0) HIP_TRY(hipEventRecord(q.ev_h2d_start, sh));
            HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToDevice, sh));
            continue;
        }
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, h_wtns[k]) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();                       // an unregistered pointer is reported as an error: not one
        const uint8_t *src = h_wtns[k];
        if (!pinned) {
            if (count == 1 && bytes >= ((size_t)4 << 20) && p->in_flight == 0) {
                // A large pageable witness (128 MiB at 2^22): pieces alternate between two upload streams, so the staging of
                // piece i+1 (host function: four threads of memcpy) runs beside the DMA of piece i.  Staging and DMA of the whole
                // vector one after the other were 5 ms on the critical path of a synchronous zk_prove: 39.4 -> 38.2 ms at 2^22.
                // Only when no other proof is in flight: in a full pipeline the upload is hidden anyway and the sixteen extra
                // stream operations cost 1 % of the period.
                const size_t npc = bytes >= ((size_t)32 << 20) ? 8 : 4, per = ((bytes / npc) + 4095) & ~(size_t)4095;      // (pieces of >= 1 MiB)
                // staging: the slot's full-size pinned copy when a pipelined submission has made one, else a ring of two pieces
                // (one per stream: piece c + 2 is staged after the DMA of piece c, which stream order guarantees) — the first
                // proof of a process (the one-shot CLI's only one) no longer waits ~20 ms for 128 MiB of pinned memory
                if (!q.wtns_pin && !q.pin_ring) HIP_TRY(hipHostMalloc((void **)&q.pin_ring, 2 * per, hipHostMallocDefault));
                if (!q.ev_h2d_b) HIP_TRY(hipEventCreateWithFlags(&q.ev_h2d_b, hipEventDisableTiming));
                hipStream_t sb = p->stream_fin;          // idle: nothing is in flight (no stream of its own: hardware queues are few)
                if (src_ready) HIP_TRY(hipStreamWaitEvent(sb, src_ready, 0));
                // stream B must not touch the slot's buffers before stream A's earlier work (the previous use of this slot) is done
                HIP_TRY(hipEventRecord(q.ev_h2d_b, sh));
                HIP_TRY(hipStreamWaitEvent(sb, q.ev_h2d_b, 0));
                if (tm) HIP_TRY(hipEventRecord(q.ev_h2d_start, sh));
                for (size_t c = 0, off = 0; off < bytes; c++, off += per) {
                    const size_t len = off + per < bytes ? per : bytes - off;
                    hipStream_t st = (c & 1) ? sb : sh;
                    uint8_t *stg = q.wtns_pin ? q.wtns_pin + off : q.pin_ring + (c & 1) * per;
                    q.stage_chunk[c] = StageJob{stg, h_wtns[k] + off, len};
                    HIP_TRY(hipLaunchHostFunc(st, stage_job_run, &q.stage_chunk[c]));
                    HIP_TRY(hipMemcpyAsync(dst + off, stg, len, hipMemcpyHostToDevice, st));
                }
                HIP_TRY(hipEventRecord(q.ev_h2d_b, sb));
                HIP_TRY(hipStreamWaitEvent(sh, q.ev_h2d_b, 0));
                continue;
            }
            if (!q.wtns_pin) HIP_TRY(hipHostMalloc((void **)&q.wtns_pin, bytes * p->batch, hipHostMallocDefault));
            q.stage[k] = StageJob{q.wtns_pin + (size_t)k * bytes, h_wtns[k], bytes};
            static const bool sync_stage = probe_env("ZKHIP_STAGE_SYNC") != nullptr;      // tuning aid: stage inside the call
            if (sync_stage) stage_job_run(&q.stage[k]);
            else HIP_TRY(hipLaunchHostFunc(sh, stage_job_run, &q.stage[k]));
            src = q.stage[k].dst;
        }
        if (tm && k == 0) HIP_TRY(hipEventRecord(q.ev_h2d_start, sh));      // ZK_T_WTNS_H2D: the DMA (of the first vector on), not the staging
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, sh));
    }
    if (count < p->batch) HIP_TRY(hipMemsetAsync((uint8_t *)q.wtns_dev.p + (size_t)count * bytes, 0, (size_t)(p->batch - count) * bytes, sh));
    HIP_TRY(hipEventRecord(q.ev_h2d, sh));
    if (tm) HIP_TRY(hipEventRecord(q.ev[12], sh));
    return q.wtns_dev.p;
}
static const Fr *upload_witness(zk_prover *p, zk_prover::ProofSlot &q, const uint8_t *h_wtns, hipEvent_t src_ready = nullptr) {
    return upload_witnesses(p, q, &h_wtns, 1, src_ready);
}

// Steps 1-10 of prove() (src/groth16.cpp:52-204), device part: everything is enqueued, nothing waits.
// The work of one proof is split into phases so that a prover holding one block of a chain that is
// PARTITIONED across GPUs (ZK_FLAG_PARTITIONED_CHAIN) can stop where the blocks have to be exchanged:
//   front      : slot, witness upload, sort(w) + MSM B2/A/B1 on stream 2, a = A.w, b = B.w, c = a o b
//   [cross DIF]: the log2(G) top stages of the three inverse transforms          (partitioned only)
//   local      : the local stages of the inverse and forward transforms, coset shift fused
//   [cross DIT]: the log2(G) top stages of the three forward transforms          (partitioned only)
//   back       : h, sort(h), MSM H and C, joins, D2H of the window sums
// An unpartitioned prover runs front + local + back back to back (submit_locked).  Exactly one of
// d_wtns (device pointer, nVars x 32 B, must stay valid until the proof is collected) and h_wtns
// (host pointer) is given.  Caller holds p->mtx.
namespace {

struct PhaseCtx {
    zk_prover *p;
    zk_prover::ProofSlot &q;
    hipStream_t s, s2, sf;
    hipStream_t tail[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int ntails = 0;
    Fr *abc, *h;
    SortBufs *sort_h;
    bool tm;
    uint32_t tbw, tbh, Ww;
    uint64_t ew, eh;
    MsmPlan pw;
    G1Acc *bA, *bB1, *bC, *bH;
    PhaseCtx(zk_prover *p_, int si) : p(p_), q(p_->slot[si]) {
        const int lane = si % p->lanes;
        if (lane == 0) {
            s = p->stream; s2 = p->stream2; sf = p->stream_fin;
            if (q.use_tails) {
                for (int m = 0; m < 5; m++) tail[m] = p->tail[m];
                ntails = p->tail_streams;
            }
            abc = p->abc_use; h = p->h.p; sort_h = &p->sort_h;
        } else {
            zk_prover::LaneExtra &x = *p->extra[lane - 1];
            x.ensure();
            s = x.stream; s2 = x.stream2; sf = x.stream;
            abc = x.abc.p; h = x.h.p; sort_h = &x.sort_h;
        }
        tm = (p->flags & ZK_FLAG_TIMINGS) != 0;
        tbw = q.sort_w.total_buckets(); tbh = p->sort_h.total_buckets();
        ew = q.sort_w.max_entries(); eh = p->sort_h.max_entries();
        pw = q.sort_w.plan; Ww = pw.sets * msm_wsum_rc(pw);      // records of one witness MSM in the window-sum arrays
        bA = q.buckets_g1.p; bB1 = bA + tbw; bC = bB1 + tbw; bH = bC + tbw;
    }
    void mark(int i) const { if (tm) HIP_TRY(hipEventRecord(q.ev[i], s)); }
    AccumTail tail_of(int m) const { AccumTail t; t.stream = tail[m]; t.l1_done = q.ev_l1[m]; t.buckets_zeroed = q.zeroed; t.chunk_min = q.l1_chunk_min; t.chunk_max = q.l1_chunk_max; return t; }
    hipStream_t after(int m, hipStream_t own) const { return tail[m] ? tail[m] : own; }
    NttTables tables() const { return NttTables{p->logn, p->tw_fwd.p, p->tw_inv.p, p->tw_coset.p, p->tw_ninv.p}; }
};


// ZKHIP_LONE_ORDER=1 (-DZK_PROBES builds only): the witness MSMs of a lone proof enqueued BEHIND the transform chain.
// Measured and left off (profiles/r04c_ab_lone_order.txt, same box, three alternations of six synchronous proofs):
// 2^22 40.4 / 40.0 / 39.7 ms with, 39.4 / 39.9 / 39.4 without; 2^20 13.4 / 13.3 / 13.8 vs 13.1 / 14.6 / 13.6.
static bool lone_order(const zk_prover *p) {
    static const int forced = [] { const char *e = probe_env("ZKHIP_LONE_ORDER"); return e ? atoi(e) : -1; }();
    return forced > 0 && p->shard_count == 1 && !p->part;
}

// MSM B2, A, B1 over the shared bucket order of sort(w), on stream 2 (src/groth16.cpp:180-197)
static void enqueue_witness_msms(zk_prover *p, PhaseCtx &c) {
    zk_prover::ProofSlot &q = c.q;
    hipStream_t s2 = c.s2;
    const bool tails = c.ntails != 0;
    const bool tm = c.tm;
    // follow-up kernels (partial merges, bucket reductions) are small and latency-bound: on their
    // own streams they neither delay the next level-1 kernel of their MSM's stream nor pile up
    // behind the last one
    const uint32_t tbw = c.tbw, Ww = c.Ww;
    const uint64_t ew = c.ew;
    const MsmPlan pw = c.pw;
    // A LONE proof (nothing else in flight: slot 0, lane 0) has an idle stream — the finishing one, which only joins and copies
    // at the very end.  The merges and the bucket reduction of MSM B2 go there instead of standing in stream 2's line: the
    // A|B1|C launch starts right behind the G2 level-1 launch, and the G2 reduction no longer ends the proof.  Medians of 16
    // synchronous proofs, three alternations (profiles/r04an_g2_aside_medians.txt): 2^14 1.61 -> 1.37 ms, 2^16 2.31 -> 1.88,
    // 2^17 2.98 -> 2.58, 2^18 4.46 -> 4.18, 2^19 7.52 -> 7.35; at 2^22 it LOST 0.2-2 ms in round 4 (r04am: the merges then run beside the
    // chip-filling A|B1|C launch, which they slow down more than their own 0.5 ms), hence a size limit.  Round 6, with the wave
    // priorities and the split reduction in place (same box, three alternations, one proof at a time with a resident witness /
    // synchronous with a host witness, profiles/r06g_*, r06h_*): 2^20 10.71 / 10.61 / 10.62 -> 10.01 / 9.93 / 9.90 ms (the G2 reduction
    // was the LAST kernel chain of such a proof: 1.0 ms behind MSM A and B1 on stream 2), 2^21 17.85 / 18.38 / 18.00 -> 17.12 / 16.95 /
    // 16.94; 2^22 and 2^24 equal within the noise (32.87 vs 32.75, 124.5 vs 123.8); pipelined periods unchanged everywhere: limit 2^21.
    static const uint32_t g2_aside_maxlog = [] { const char *e = probe_env("ZKHIP_G2_ASIDE_MAXLOG"); return e ? (uint32_t)atoi(e) : 21u; }();
    const bool g2_aside = !tails && c.sf != s2 && c.sf != c.s && p->in_flight == 0 && !p->capturing && !p->use_graph &&
                          p->sv.size() <= ((uint64_t)1 << g2_aside_maxlog);      // (the witness slice: a shard of a large proof is a small MSM)
    // (The WHOLE MSM B2 there, its level-1 launch beside the A|B1|C one, was measured too: nothing at 2^14 ... 2^16, +4-7 % at
    // 2^17 / 2^18, profiles/r04ap_g2_whole_aside.txt.)
    AccumTail t4 = c.tail_of(4);
    if (g2_aside) t4.stream = c.sf;
    launch_msm_accum_g2(q.buckets_g2.p, q.sort_w.offsets.p, q.sort_w.entries.p, p->ptsB2.p, 0, 0, tbw, ew, q.acc_ws_g2.p, q.acc_key[4], q.acc_flag[4], s2, tm ? &q.ev[10] : nullptr, t4);
    if (tails) launch_msm_reduce_g2(q.wsum_g2.p, q.scratch_g2.p, q.buckets_g2.p, 1, pw, c.tail[4]);
    else if (g2_aside) launch_msm_reduce_g2(q.wsum_g2.p, q.scratch_g2.p, q.buckets_g2.p, 1, pw, c.sf);
    if (p->batch_abc) {
        // small circuits: MSM A, B1 and C (same scalars, same sorted entries) in ONE set of launches —
        // level-1 accumulation, merges and bucket reduction each cost what one MSM's cost
        AccumBatch b;
        memset(&b, 0, sizeof b);
        b.n = 3;
        b.points[0] = p->ptsA.p; b.points[1] = p->ptsB1.p; b.points[2] = p->ptsC.p;
        // ZKHIP_PROBE_ONE_TABLE=1 (-DZK_PROBES builds only; wrong sums): all three MSMs of the launch gather from table A — the same
        // instruction stream and gather count out of a 3.25 GiB footprint instead of 9.75 GiB (profiles/NEGATIVE_RESULTS.md item 18)
        static const bool one_table = probe_env("ZKHIP_PROBE_ONE_TABLE") != nullptr;
        if (one_table) b.points[1] = b.points[2] = p->ptsA.p;
        b.idx_min[2] = b.idx_sub[2] = p->c_idx_min;
        b.bucket_stride = tbw;
        b.ws_stride = q.acc_stride;
        launch_msm_accum_g1_batch(c.bA, q.sort_w.offsets.p, q.sort_w.entries.p, b, tbw, ew, q.acc_ws_g1[0], q.acc_key[0], q.acc_flag[0], s2, tm ? &q.ev[8] : nullptr, c.tail_of(0));
        if (tm) for (int e : {13, 14, 15, 16}) HIP_TRY(hipEventRecord(q.ev[e], s2));      // (B1 and C have no launch of their own)
        launch_msm_reduce_g1(q.wsum_g1.p, q.scratch_g1.p, c.bA, 3, pw, c.after(0, s2));
        if (!tails && !g2_aside) launch_msm_reduce_g2(q.wsum_g2.p, q.scratch_g2.p, q.buckets_g2.p, 1, pw, s2);
        HIP_TRY(hipEventRecord(q.ev_join, s2));
    } else {
    launch_msm_accum_g1(c.bA, q.sort_w.offsets.p, q.sort_w.entries.p, p->ptsA.p, 0, 0, tbw, ew, q.acc_ws_g1[0], q.acc_key[0], q.acc_flag[0], s2, tm ? &q.ev[8] : nullptr, c.tail_of(0));
    if (tails) launch_msm_reduce_g1(q.wsum_g1.p, q.scratch_g1.p, c.bA, 1, pw, c.tail[0]);
    launch_msm_accum_g1(c.bB1, q.sort_w.offsets.p, q.sort_w.entries.p, p->ptsB1.p, 0, 0, tbw, ew, q.acc_ws_g1[1], q.acc_key[1], q.acc_flag[1], s2, tm ? &q.ev[13] : nullptr, c.tail_of(1));
    if (tails) {
        launch_msm_reduce_g1(q.wsum_g1.p + Ww, q.scratch_g1.p + msm_reduce_scratch_points(1, pw), c.bB1, 1, pw, c.tail[1]);
    } else {
        // bucket reductions stay on the stream of their MSMs
        if (!g2_aside) launch_msm_reduce_g2(q.wsum_g2.p, q.scratch_g2.p, q.buckets_g2.p, 1, pw, s2);
        launch_msm_reduce_g1(q.wsum_g1.p, q.scratch_g1.p, c.bA, 2, pw, s2);
    }
    HIP_TRY(hipEventRecord(q.ev_join, s2));
    }

}

}   // namespace

namespace zkp {

int phase_front(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, const uint8_t *r32, const uint8_t *s32, hipEvent_t src_ready, const BatchIn *bi) {
    DeviceGuard g(p->device);
    if (p->in_flight >= p->ring) throw std::invalid_argument("too many proofs in flight (ZK_MAX_IN_FLIGHT, or the depth given to zk_prover_reserve): collect one first");
    if (p->phase_open >= 0) throw std::invalid_argument("a proof is still being submitted phase by phase");
    // nothing in flight: start over at slot 0 — a caller that proves one at a time (zk_prove, the reference's contract) then
    // lives in ONE slot on lane 0 instead of walking through all eight (each with ~1 GiB of buffers at 2^22, a pinned
    // staging copy and, per lane, two streams its first use has to create: the first eight proofs of a prover paid
    // 10-20 ms each for that).  (Not while a graph is being captured: submit_graph has chosen the slot.)
    if (p->in_flight == 0 && !p->capturing) p->next_submit = p->next_collect = 0;
    const int si = (int)(p->next_submit % p->ring);
    alloc_slot(p, si);
    // Follow-up streams only for a proof that is submitted while another one is in flight: they let the next level-1 launch
    // start beside the merges and reductions of the previous MSM — worth 2 % of a sharded prover's period with two in flight —
    // but every hop to them is an event across hardware queues, and a LONE proof is faster without (same box, with / without:
    // 2^22 synchronous 39.0-39.8 / 37.9-38.0 ms, 2^20 13.3 / 12.4 ms; rank-0 share of 8 shards one at a time 7.8-8.0 / 6.5 ms).
    // (A captured graph keeps whatever its slot recorded.)
    if (!p->capturing) p->slot[si].use_tails = p->in_flight > 0 || p->use_graph;
    // Entries per level-1 lane.  A LONE proof wants every lane the chip holds (32 entries each at least); a proof submitted beside
    // others shares the chip anyway, and fewer, longer lanes leave fewer cut runs to merge: with 128 at least the pipelined period
    // is 2^17 -1.5 %, 2^18 -7 %, 2^19 -2.5 %, 2^20 -1.8 %, 2^22 with a realistic witness -3 ... -4 % (its witness MSMs are sparse), 2^16 +16 % (left out) — and one proof
    // at a time +0.3 ... 0.4 ms had it been applied there (profiles/r05zk_*, r05zl_*, r05zm_*).
    if (!p->capturing) {
        static const uint32_t thr_min = [] { const char *e = probe_env("ZKHIP_L1_CHUNK_MIN_BUSY"); return e ? (uint32_t)atoi(e) : 128u; }();
        static const uint32_t thr_log = [] { const char *e = probe_env("ZKHIP_L1_CHUNK_MIN_BUSY_LOG"); return e ? (uint32_t)atoi(e) : 17u; }();
        static const uint32_t thr_max = [] { const char *e = probe_env("ZKHIP_L1_CHUNK_MAX_BUSY"); return e ? (uint32_t)atoi(e) : 1280u; }();
        // the shards of a sharded proof too, by the size of a shard: rank-0 share of eight shards of 2^22 with two in flight 5.42 -> 5.00 ms,
        // of 2^24 16.65 -> 16.30, two shards 16.6 -> 16.3, four equal (profiles/r05zt_busy_lanes_shards.txt; ZKHIP_L1_BUSY_SHARDS=0: probe)
        static const bool shards_too = [] { const char *e = probe_env("ZKHIP_L1_BUSY_SHARDS"); return e ? atoi(e) != 0 : true; }();
        uint32_t lg_sh = 0;
        while ((1u << (lg_sh + 1)) <= p->shard_count) lg_sh++;
        const bool busy = p->in_flight > 0 && !p->use_graph && ((p->shard_count == 1 && !p->part) || shards_too) && p->batch == 1 && p->logn >= thr_log + lg_sh;
        p->slot[si].l1_chunk_min = busy ? thr_min : 0u;
        p->slot[si].l1_chunk_max = busy ? thr_max : 0u;      // fewer rounds of lanes: at 2^22 H runs as one round of 277 entries per lane, A|B1|C as one of 832
                                                             // (160 / 320 / 640 / 1280 / one round always: 2^22 30.65 / 30.59 / 30.29 / 30.07 / 30.08 ms, 2^24 121.4 / 118.8 / 116.4 / 116.1 / 118.4; profiles/r05zn_*, r05zo_*)
    }
    PhaseCtx c(p, si);
    zk_prover::ProofSlot &q = c.q;
    if (!p->capturing) p->launches_at_front = g_kernel_launches.load(std::memory_order_relaxed);
    bool staged = h_wtns != nullptr;
    q.count = 1;
    if (bi) {
        if (bi->count < 1 || bi->count > p->batch) throw std::invalid_argument("batch submission: between 1 and opts.batch witnesses");
        d_wtns = upload_witnesses(p, q, bi->wtns, bi->count, src_ready);
        staged = true;
        q.count = bi->count;
        r32 = bi->r32s;
        s32 = bi->s32s;
    } else if (staged) {
        d_wtns = upload_witness(p, q, h_wtns, src_ready);
    } else if (p->batch > 1) {             // a device witness on a batch prover: vector 0 of the slot's buffer, the rest zero
        d_wtns = upload_witnesses(p, q, nullptr, 1, src_ready, d_wtns);
        staged = true;
    }
    q.host_witness = h_wtns != nullptr || bi != nullptr;
    q.have_r = r32 != nullptr;
    q.have_s = s32 != nullptr;
    if (r32) memcpy(q.r32, r32, (size_t)32 * q.count);
    if (s32) memcpy(q.s32, s32, (size_t)32 * q.count);
    hipStream_t s = c.s, s2 = c.s2;
    const bool tails = c.ntails != 0;
    const bool tm = c.tm;

    c.mark(0);
    // ---- stream2: work that depends on the witness only (the reference runs it AFTER the FFT
    // chain, src/groth16.cpp:180-204; it is independent of it): sort(w) once, then MSM B2, A, B1
    // over the shared bucket order.  MSM C joins stream 1 behind MSM H to balance the streams.
    // stream2 depends on stream 1 only through a staged witness: with a caller-owned device witness
    // it runs ahead, so that proof k+1's witness MSMs follow proof k's directly instead of waiting
    // for proof k's stream-1 work (at 2^20 that wait left stream2 idle for a third of the period)
    // The five bucket arrays are cleared HERE, in front of the wait for the witness: the 0.4 GiB of memsets run while the
    // upload is still on its way (2.4 ms of PCIe time in which a lone proof has nothing else to do) instead of in front of
    // every level-1 launch (where they showed up as 0.5-0.6 ms each beside another proof-filling kernel,
    // profiles/r04b_lone_proof_timeline_2p22.txt).  The slot's previous proof has been collected: nothing reads them.
    q.zeroed = !p->capturing && !p->use_graph;
    if (q.zeroed) {
        HIP_TRY(hipMemsetAsync(c.bA, 0, (size_t)3 * c.tbw * sizeof(G1Acc), s2));        // A | B1 | C (C's launch on stream 1 waits for sort(w) on stream 2)
        HIP_TRY(hipMemsetAsync(q.buckets_g2.p, 0, (size_t)c.tbw * sizeof(G2Acc), s2));
        HIP_TRY(hipMemsetAsync(c.bH, 0, (size_t)c.tbh * sizeof(G1Acc), s));
    }
    if (staged) {
        HIP_TRY(hipStreamWaitEvent(s, q.ev_h2d, 0));
        HIP_TRY(hipStreamWaitEvent(s2, q.ev_h2d, 0));
    }
    if (p->capturing && s2 != s) {       // stream 2 joins the capture (and the graph orders it behind the upload)
        HIP_TRY(hipEventRecord(q.ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(s2, q.ev_fork, 0));
    }
    q.sort_w.run(d_wtns + p->sv.lo, s2);
    HIP_TRY(hipEventRecord(q.ev_sortw, s2));
    // Experiment (off: lone_order): the witness MSMs of a LONE proof enqueued behind the transform chain.  The idea: the G2
    // accumulation is ONE round of workgroups that hold every register of the chip for its whole 11 ms — started beside the
    // chain it starves the chain's last pass (0.9 ms of work took 16.6 ms, profiles/r04a_lone_proof_timeline_2p22.txt), so h,
    // sort(h) and MSM H only begin when MSM A is done.  Behind the chain the starved kernel is sort(h)'s partition pass instead
    // (9.4 ms beside the G2 launch, profiles/r04b_lone_proof_timeline_2p22.txt) and the proof is no shorter: a lone proof is
    // bound by the SUM of its chip-filling kernels, not by their order.  Re-measured at the end of round 4
    // with the wave priorities in place, A|B1|C batched, and the sort's workgroups cut to 512 / 256 threads so that they fit
    // beside a level-1 launch's waves (profiles/r04bq_lone_order_sort_workgroups.txt): 36.8 ms without, 37.0-37.4 with, and the
    // smaller sort workgroups cost 0.7-1.5 ms by themselves.
    q.defer_w = lone_order(p) && !p->capturing && !p->use_graph && p->in_flight == 0 && s2 != s;
    if (!q.defer_w) enqueue_witness_msms(p, c);

    // ---- stream: the h chain (LDS/latency-bound passes overlap with the MSMs above)
    // 1-3: a = A.w, b = B.w, c = a o b   (src/groth16.cpp:52-96) — on the rows this prover holds
    const uint64_t nl = p->nloc;
    Fr *abc = c.abc;
    CsrDev csr{p->csr_rowptr.p, p->csr_col.p, p->csr_val.p};
    // (a batch: vector v's a|b|c behind vector v-1's; unused vectors are skipped)
    launch_spmv_abc(abc, abc + nl, abc + 2 * nl, csr, d_wtns, (uint32_t)nl, s, q.count, 3 * nl, p->nVars);
    c.mark(1);
    if (p->part && !p->have_peers && p->pk_use) launch_chunk_pack(p->pk_use, abc, 3, p->logn, p->log_shards, s);   // -> all_to_all #1
    p->phase_open = si;
    p->phase_next = p->part ? 1 : 2;
    return si;
}

// The log2(G) stages over the top index bits of the three transforms (ntt.hip, launch_ntt_cross).
// Inside one process (peer buffers known) this prover first pushes chunk s of its block into GPU s's
// exchange buffer and records `pushed`; the caller makes every prover wait for every other's event
// (cross_wait) before cross_run.  Between processes the caller has done the all-to-all itself.
void phase_cross_push(zk_prover *p, hipEvent_t pushed) {
    DeviceGuard g(p->device);
    launch_chunk_scatter(p->peer_xb, p->abc_use, 3, p->logn, p->log_shards, p->shard_index, p->stream);
    HIP_TRY(hipEventRecord(pushed, p->stream));
}
void phase_cross_run(zk_prover *p, bool inverse, hipEvent_t done) {
    if (p->phase_open < 0 || p->phase_next != (inverse ? 1 : 3)) throw std::invalid_argument("chain phases out of order");
    DeviceGuard g(p->device);
    PhaseCtx c(p, p->phase_open);
    const uint64_t nl = p->nloc, chunk = nl >> p->log_shards;
    if (p->have_peers) {
        // exchange buffer [poly][source GPU][chunk]; results go straight into the owners' blocks (peer writes)
        launch_ntt_cross(inverse, p->xb_use, chunk, nl, p->peer_abc, nl, (uint64_t)p->shard_index * chunk, 3, c.tables(), p->log_shards, p->shard_index, c.s);
    } else {
        // exchange buffer [source GPU][poly][chunk] (one all_to_all_single delivered it); results in place
        if (!p->pk_use) throw std::invalid_argument("no exchange buffers registered (zk_shard_set_exchange)");
        Fr *inplace[8];
        for (uint32_t i = 0; i < 8; i++) inplace[i] = p->xb_use + (uint64_t)i * 3 * chunk;
        launch_ntt_cross(inverse, p->xb_use, 3 * chunk, chunk, inplace, chunk, 0, 3, c.tables(), p->log_shards, p->shard_index, c.s);
    }
    if (done) HIP_TRY(hipEventRecord(done, c.s));
    p->phase_next = inverse ? 2 : 4;
}

void phase_local(zk_prover *p) {
    if (p->phase_open < 0 || p->phase_next != 2) throw std::invalid_argument("chain phases out of order");
    DeviceGuard g(p->device);
    PhaseCtx c(p, p->phase_open);
    // 4: three coset evaluations (src/groth16.cpp:98-155), batched, no bit-reversal pass; on a partitioned
    // chain only the stages over the low logn - log2(G) index bits of this prover's block
    const uint32_t local_logn = p->logn - p->log_shards_chain();
    const uint64_t nl = p->nloc;
    NttTables tb = c.tables();
    const bool a2a = p->part && !p->have_peers;          // blocks travel through the caller's all_to_all
    if (a2a) launch_chunk_unpack(c.abc, p->pk_use, 3, p->logn, p->log_shards, c.s);
    if (p->pair.L) {
        // ZKHIP_PROBE_SKIP_NTT (-DZK_PROBES builds only, WRONG results): only a is transformed, b and c are not — two thirds of the
        // transforms made free (h stays a dense vector of full-size scalars; with none transformed a.b - c would be 0)
        static const bool skip_ntt = probe_env("ZKHIP_PROBE_SKIP_NTT") != nullptr;
        launch_ntt_coset_pair(c.abc, nl, (skip_ntt ? 1 : 3) * c.q.count, p->pair, c.s);      // inverse, coset shift * 1/n, forward: nttpair.hip
    } else {
        launch_ntt_dif_inverse(c.abc, nl, 3 * c.q.count, tb, c.s, local_logn);
        launch_ntt_dit_forward(c.abc, nl, 3 * c.q.count, tb, c.s, p->tw_coset.p + (p->part ? p->sh.lo : 0), local_logn);   // coset shift * 1/n fused into the first pass
    }
    if (a2a) launch_chunk_pack(p->pk_use, c.abc, 3, p->logn, p->log_shards, c.s);
    if (c.q.defer_w) {                   // lone proof: now the witness MSMs (phase_front)
        HIP_TRY(hipEventRecord(c.q.ev_chain, c.s));
        HIP_TRY(hipStreamWaitEvent(c.s2, c.q.ev_chain, 0));
        enqueue_witness_msms(p, c);
        c.q.defer_w = false;
    }
    p->phase_next = p->part ? 3 : 4;
}

void phase_back(zk_prover *p) {
    if (p->phase_open < 0 || p->phase_next != 4) throw std::invalid_argument("chain phases out of order");
    DeviceGuard g(p->device);
    PhaseCtx c(p, p->phase_open);
    zk_prover::ProofSlot &q = c.q;
    hipStream_t s = c.s;
    const uint64_t nl = p->nloc;
    Fr *abc = c.abc;
    if (p->part && !p->have_peers) launch_chunk_unpack(abc, p->pk_use, 3, p->logn, p->log_shards, s);
    // 5: h = fromMontgomery(a.b - c)  (src/groth16.cpp:157-163)
    launch_abc_to_h(c.h, abc, abc + nl, abc + 2 * nl, nl, s, q.count, 3 * nl);
    if (q.count < p->batch) HIP_TRY(hipMemsetAsync(c.h + nl * q.count, 0, (size_t)(p->batch - q.count) * nl * sizeof(Fr), s));    // h of an unused vector: no digits
    c.mark(2);
    c.sort_h->run(c.h + (p->part ? 0 : p->sh.lo), s);
    c.mark(3);
    // 6: MSM H (src/groth16.cpp:171-173) and its bucket reduction
    launch_msm_accum_g1(c.bH, c.sort_h->offsets.p, c.sort_h->entries.p, p->ptsH.p, 0, 0, c.tbh, c.eh, q.acc_ws_g1[3], q.acc_key[3], q.acc_flag[3], s, c.tm ? &q.ev[17] : nullptr, c.tail_of(3));
    c.mark(4);
    launch_msm_reduce_g1(q.wsum_g1.p + 3 * c.Ww, q.scratch_g1.p + msm_reduce_scratch_points(3, c.pw), c.bH, 1, p->sort_h.plan, c.after(3, s));
    // MSM C (src/groth16.cpp:202-204) balances the two streams: it only needs sort(w).  (Moving it to
    // stream2 was measured slower at every shard count; so was raising stream 1's priority for
    // anything but the two-in-flight throughput of 4-8 shards.)
    if (!p->batch_abc) {
    HIP_TRY(hipStreamWaitEvent(s, q.ev_sortw, 0));
    launch_msm_accum_g1(c.bC, q.sort_w.offsets.p, q.sort_w.entries.p, p->ptsC.p, p->c_idx_min, p->c_idx_min, c.tbw, c.ew, q.acc_ws_g1[2], q.acc_key[2], q.acc_flag[2], s, c.tm ? &q.ev[15] : nullptr, c.tail_of(2));
    launch_msm_reduce_g1(q.wsum_g1.p + 2 * c.Ww, q.scratch_g1.p + msm_reduce_scratch_points(2, c.pw), c.bC, 1, c.pw, c.after(2, s));
    }
    c.mark(5);
    HIP_TRY(hipEventRecord(q.ev_main, s));

    // ---- join on the finishing stream (the main streams go straight on to the next proof):
    // window sums -> pinned host memory
    hipStream_t sf = c.sf;
    HIP_TRY(hipStreamWaitEvent(sf, q.ev_main, 0));
    HIP_TRY(hipStreamWaitEvent(sf, q.ev_join, 0));
    for (int i = 0; i < c.ntails; i++) {
        HIP_TRY(hipEventRecord(q.ev_tail[i], p->tail_pool[i]));
        HIP_TRY(hipStreamWaitEvent(sf, q.ev_tail[i], 0));
    }
    if (c.tm) HIP_TRY(hipEventRecord(q.ev[6], sf));
    HIP_TRY(hipMemcpyAsync(q.w1, q.wsum_g1.p, q.w1_bytes, hipMemcpyDeviceToHost, sf));
    HIP_TRY(hipMemcpyAsync(q.w2, q.wsum_g2.p, q.w2_bytes, hipMemcpyDeviceToHost, sf));
    HIP_TRY(hipEventRecord(q.ev_done, sf));
    if (p->capturing && sf != s) HIP_TRY(hipStreamWaitEvent(s, q.ev_done, 0));     // every forked stream rejoins the origin
    HIP_TRY(hipGetLastError());          // nothing of the ~100 launches above may have been refused
    // kernel launches of this proof (zk_prover_info): exact while one thread submits at a time in the process
    if (!p->capturing) p->launches_last_proof = g_kernel_launches.load(std::memory_order_relaxed) - p->launches_at_front;
    q.busy = true;
    q.via_graph = false;
    p->phase_open = -1;
    p->next_submit++;
    p->in_flight++;
}


}   // namespace zkp

// Graph path (small circuits): a proof's device work — everything behind the witness upload — is recorded once
// per slot by running the same three phases under stream capture, then replayed with one hipGraphLaunch on the
// lane's stream 1.  The upload stays outside (its source changes with every proof), and so does the event a
// collect waits for.
static void submit_graph(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, const uint8_t *r32, const uint8_t *s32) {
    if (p->in_flight >= p->ring) throw std::invalid_argument("too many proofs in flight (ZK_MAX_IN_FLIGHT, or the depth given to zk_prover_reserve): collect one first");
    if (p->phase_open >= 0) throw std::invalid_argument("a proof is still being submitted phase by phase");
    if (p->in_flight == 0) p->next_submit = p->next_collect = 0;      // (see phase_front)
    const int si = (int)(p->next_submit % p->ring);
    alloc_slot(p, si);
    PhaseCtx c(p, si);
    zk_prover::ProofSlot &q = c.q;
    if (!q.ev_gdone) HIP_TRY(hipEventCreateWithFlags(&q.ev_gdone, hipEventDisableTiming));
    const bool staged = h_wtns != nullptr;
    if (staged) d_wtns = upload_witness(p, q, h_wtns);
    if (!q.gexec || q.graph_wtns != d_wtns) {
        if (q.gexec) { (void)hipGraphExecDestroy(q.gexec); q.gexec = nullptr; }
        if (q.graph) { (void)hipGraphDestroy(q.graph); q.graph = nullptr; }
        const uint32_t ns = p->next_submit, nf = p->in_flight;
        HIP_TRY(hipStreamBeginCapture(c.s, hipStreamCaptureModeRelaxed));
        p->capturing = true;
        hipError_t end = hipSuccess;
        try {
            PhaseAbort guard{p};
            phase_front(p, d_wtns, nullptr, r32, s32);
            phase_local(p);
            phase_back(p);
            guard.armed = false;
        } catch (...) {
            p->capturing = false;
            hipGraph_t dead = nullptr;
            (void)hipStreamEndCapture(c.s, &dead);
            if (dead) (void)hipGraphDestroy(dead);
            p->next_submit = ns; p->in_flight = nf; q.busy = false;
            throw;
        }
        p->capturing = false;
        end = hipStreamEndCapture(c.s, &q.graph);
        p->next_submit = ns; p->in_flight = nf; q.busy = false;      // the capture ran the bookkeeping of a submit: undo
        HIP_TRY(end);
        HIP_TRY(hipGraphInstantiate(&q.gexec, q.graph, nullptr, nullptr, 0));
        q.graph_wtns = d_wtns;
    }
    q.host_witness = staged;
    q.have_r = r32 != nullptr;
    q.have_s = s32 != nullptr;
    if (r32) memcpy(q.r32, r32, 32);
    if (s32) memcpy(q.s32, s32, 32);
    if (staged) HIP_TRY(hipStreamWaitEvent(c.s, q.ev_h2d, 0));
    HIP_TRY(hipGraphLaunch(q.gexec, c.s));
    HIP_TRY(hipEventRecord(q.ev_gdone, c.s));
    q.busy = true;
    q.via_graph = true;
    p->next_submit++;
    p->in_flight++;
}

static void submit_locked(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, const uint8_t *r32, const uint8_t *s32, const BatchIn *bi = nullptr) {
    if (p->part) throw std::invalid_argument("this prover holds one block of a partitioned chain: drive it through zk_multi_prove* or zk_shard_*");
    DeviceGuard g(p->device);
    if (p->use_graph) {
        submit_graph(p, d_wtns, h_wtns, r32, s32);
        return;
    }
    PhaseAbort guard{p};
    phase_front(p, d_wtns, h_wtns, r32, s32, nullptr, bi);
    phase_local(p);
    phase_back(p);
    guard.armed = false;
}

// Waits for the oldest proof in flight, then the host part: Horner over the window sums
// (c doublings per window).  Takes p->cmtx for the whole call and p->mtx only around the bookkeeping:
// submissions go on while this thread waits and computes.  rs (optional) receives the proof's (r, s).
void zkp::collect_sums(zk_prover *p, zk_msm_sums *out, SubmittedRS *rs, const DirectProof *direct) {
    std::lock_guard<std::mutex> ck(p->cmtx);
    zk_prover::ProofSlot *qp;
    {
        std::lock_guard<std::mutex> lk(p->mtx);
        if (!p->in_flight) throw std::invalid_argument("no proof in flight");
        qp = &p->slot[p->next_collect % p->ring];
        // a call that cannot take THIS submission is refused BEFORE anything is retired: the submission stays the oldest
        // one and the caller's FIFO stays in step (a wrong count used to drop every proof of the submission)
        if (p->batch > 1) {
            if (!direct) throw std::invalid_argument("partial sums are not available from a batch prover");
            if (direct->count != qp->count) throw std::invalid_argument("this submission carries a different number of proofs");
        } else if (direct && direct->count != 1) {
            throw std::invalid_argument("this prover was not created for batched submissions");
        }
    }
    zk_prover::ProofSlot &q = *qp;
    DeviceGuard g(p->device);
    // What the tail needs of (r, s) alone — four fixed-base multiplications of delta, 0.13 of the tail's 0.3 ms — is done HERE,
    // before the wait: a synchronous zk_prove spends it while the GPU works instead of behind it (2^14: 1.37 ms of which 0.3 tail).
    HostTail::RsPart rs_pre[ZK_MAX_BATCH];
    int rs_failed = 0;
    if (direct) {
        std::atomic<int> failed{0};
        tail_pool().for_each(p->batch > 1 ? q.count : 1u, [&](uint32_t k) {
            const uint8_t *r32 = direct->use_submitted ? (q.have_r ? q.r32[k] : nullptr) : direct->r32;
            const uint8_t *s32 = direct->use_submitted ? (q.have_s ? q.s32[k] : nullptr) : direct->s32;
            try {
                if (HostTail::prepare_rs(p->vk_delta1, p->vk_delta2, r32, s32, &rs_pre[k])) failed.store(1);
            } catch (...) {
                failed.store(2);
            }
        });
        rs_failed = failed.load();
    }
    const hipError_t done = hipEventSynchronize(q.via_graph ? q.ev_gdone : q.ev_done);
    // the slot is retired whatever happens below (a failed proof must not wedge the queue), but only
    // AFTER the wait and the host tail: nobody may reuse its buffers while they are still read
    struct Retire {
        zk_prover *p;
        zk_prover::ProofSlot &q;
        ~Retire() {
            std::lock_guard<std::mutex> lk(p->mtx);
            p->next_collect++;
            p->in_flight--;
            q.busy = false;
        }
    } retire{p, q};
    HIP_TRY(done);
    if (rs_failed) throw std::runtime_error(rs_failed == 1 ? "getrandom failed" : "host tail of a proof failed");
    if (rs) {
        rs->have_r = q.have_r;
        rs->have_s = q.have_s;
        memcpy(rs->r32, q.r32[0], 32);
        memcpy(rs->s32, q.s32[0], 32);
    }
    const bool tm = (p->flags & ZK_FLAG_TIMINGS) != 0;
    if (tm) {
        float ms[7], g1 = 0, g2 = 0;
        for (int i = 0; i < 5; i++) HIP_TRY(hipEventElapsedTime(&ms[i], q.ev[i], q.ev[i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms[5], q.ev[5], q.ev[6]));
        HIP_TRY(hipEventElapsedTime(&ms[6], q.ev[0], q.ev[6]));
        for (int a : {8, 13, 15, 17}) {           // the four G1 level-1 launches of this proof: mean
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, q.ev[a], q.ev[a + 1]));
            g1 += t / 4;
        }
        HIP_TRY(hipEventElapsedTime(&g2, q.ev[10], q.ev[11]));
        p->timings[ZK_T_SPMV] = ms[0];
        p->timings[ZK_T_NTT] = ms[1];                // wall time on stream 1 (shares the GPU with stream2's MSMs)
        p->timings[ZK_T_DIGITS_SORT] = ms[2];        // sort(h)
        p->timings[ZK_T_MSM_H] = ms[3];              // level-1 accumulation of MSM H on stream 1
        p->timings[ZK_T_MSM_REDUCE] = ms[4];         // MSM C on stream 1 (+ the follow-ups of H and C without follow-up streams)
        p->timings[ZK_T_JOIN_WAIT] = ms[5];          // end of stream 1's work -> every stream of the proof joined
        p->timings[ZK_T_TOTAL_DEVICE] = ms[6];
        p->timings[ZK_T_G1_L1_KERNEL] = g1;          // k_msm_accum_l1<Fq>: mean of the launches of MSM A, B1, C, H, tight events
        p->timings[ZK_T_G2_L1_KERNEL] = g2;          // k_msm_accum_l1<Fq2> of MSM B2, tight events
        float h2d = 0;
        if (q.host_witness) HIP_TRY(hipEventElapsedTime(&h2d, q.ev_h2d_start, q.ev[12]));
        p->timings[ZK_T_WTNS_H2D] = h2d;             // witness upload (own stream; 0 for device-witness proofs)
    }
    const uint32_t Ww = q.sort_w.plan.sets, Wh = p->sort_h.plan.sets;
    const uint32_t rcw = msm_wsum_rc(q.sort_w.plan), rch = msm_wsum_rc(p->sort_h.plan);
    const size_t M1 = (size_t)Ww * rcw * sizeof(G1XYZZ);          // one witness MSM's records
    const uint32_t cw = q.sort_w.plan.c, ch = p->sort_h.plan.c;
    const size_t P1 = sizeof(G1XYZZ);
    const uint8_t *w1 = q.w1, *w2 = q.w2;
    if (direct && p->batch > 1) {
        // one bucket set per proof of the submission: records [msm][proof][rc]  (count checked before the wait)
        const size_t Rw = (size_t)rcw * sizeof(G1XYZZ), Rh = (size_t)rch * sizeof(G1XYZZ), R2 = (size_t)rcw * sizeof(G2XYZZ);
        std::atomic<int> failed{0};
        tail_pool().for_each(q.count, [&](uint32_t k) {
            try {
                if (HostTail::finish_from_records(p->vk_alpha1, p->vk_beta1, p->vk_beta2, p->vk_delta1, p->vk_delta2,
                                                  w1 + k * Rw, w1 + M1 + k * Rw, w1 + 2 * M1 + k * Rw, w1 + 3 * M1 + k * Rh, w2 + k * R2,
                                                  1, cw, rcw, 1, ch, rch, nullptr, nullptr, direct->out[k].A, direct->out[k].B, direct->out[k].C,
                                                  &rs_pre[k]))
                    failed.store(1);
            } catch (...) {
                failed.store(2);
            }
        });
        if (failed.load()) throw std::runtime_error(failed.load() == 1 ? "getrandom failed" : "host tail of a batched proof failed");
        return;
    }
    if (direct) {
        if (HostTail::finish_from_windows(p->vk_alpha1, p->vk_beta1, p->vk_beta2, p->vk_delta1, p->vk_delta2, w1, w2, Ww, cw, rcw, Wh, ch, rch,
                                          nullptr, nullptr, direct->out->A, direct->out->B, direct->out->C, &rs_pre[0]))
            throw std::runtime_error("getrandom failed");
        return;
    }
    if (Ww <= 2 && Wh <= 2) {          // window-precomputed tables: one sum per MSM (two with rows for every second window), nothing to run in parallel
        HostTail::combine_windows_g1(w1, Ww, cw, rcw, out->pi_a);
        HostTail::combine_windows_g1(w1 + M1, Ww, cw, rcw, out->pib1);
        HostTail::combine_windows_g1(w1 + 2 * M1, Ww, cw, rcw, out->pi_c);
        HostTail::combine_windows_g1(w1 + 3 * M1, Wh, ch, rch, out->pih);
        HostTail::combine_windows_g2(w2, Ww, cw, rcw, out->pi_b);
        return;
    }
    // five independent serial chains (W*c doublings each): one host thread per chain
    std::thread t1([&] { HostTail::combine_windows_g1(w1, Ww, cw, rcw, out->pi_a); });
    std::thread t2([&] { HostTail::combine_windows_g1(w1 + M1, Ww, cw, rcw, out->pib1); });
    std::thread t3([&] { HostTail::combine_windows_g1(w1 + 2 * M1, Ww, cw, rcw, out->pi_c); });
    std::thread t4([&] { HostTail::combine_windows_g1(w1 + 3 * M1, Wh, ch, rch, out->pih); });
    HostTail::combine_windows_g2(w2, Ww, cw, rcw, out->pi_b);
    t1.join();
    t2.join();
    t3.join();
    t4.join();
}

// One synchronous proof.  The witness upload, the device work and the wait all happen under the
// prover's mutex: concurrent callers are serialised proof by proof (Prover::prove is re-entrant
// in the reference; here the per-proof buffers are the prover's).
static void prove_msm(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, zk_msm_sums *out, const DirectProof *direct = nullptr) {
    std::lock_guard<std::mutex> one(p->sync_mtx);
    {
        std::lock_guard<std::mutex> lk(p->mtx);
        if (p->in_flight) throw std::invalid_argument("asynchronous proofs in flight: collect them first");
        submit_locked(p, d_wtns, h_wtns, nullptr, nullptr);
    }
    collect_sums(p, out, nullptr, direct);
}

void zkp::prove_finish(zk_prover *p, const zk_msm_sums *parts, uint32_t nparts, const uint8_t *r32, const uint8_t *s32, zk_proof *out) {
    if (zk_assemble(p->vk_alpha1, p->vk_beta1, p->vk_beta2, p->vk_delta1, p->vk_delta2, parts, nparts, r32, s32, out))
        throw std::runtime_error(get_error());
}

extern "C" {

int zk_prove_msm_dev(zk_prover *p, const void *d_wtns, zk_msm_sums *partial) {
    return guarded([&] {
        if (!p || !d_wtns || !partial) throw std::invalid_argument("null argument");
        prove_msm(p, (const Fr *)d_wtns, nullptr, partial);
    });
}

int zk_prove_msm(zk_prover *p, const uint8_t *wtns, zk_msm_sums *partial) {
    return guarded([&] {
        if (!p || !wtns || !partial) throw std::invalid_argument("null argument");
        prove_msm(p, nullptr, wtns, partial);
    });
}

int zk_prove_finish(zk_prover *p, const zk_msm_sums *partials, uint32_t n_partials, const uint8_t *r32, const uint8_t *s32,
                    zk_proof *out) {
    return guarded([&] {
        if (!p || !partials || !out) throw std::invalid_argument("null argument");
        prove_finish(p, partials, n_partials, r32, s32, out);
    });
}

int zk_prove_dev(zk_prover *p, const void *d_wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out) {
    return guarded([&] {
        if (!p || !d_wtns || !out) throw std::invalid_argument("null argument");
        if (p->shard_count != 1) throw std::invalid_argument("zk_prove on a sharded prover: use zk_prove_msm + zk_prove_finish");
        const DirectProof d{out, r32, s32, false};
        prove_msm(p, (const Fr *)d_wtns, nullptr, nullptr, &d);
    });
}

int zk_prove(zk_prover *p, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out) {
    return guarded([&] {
        if (!p || !wtns || !out) throw std::invalid_argument("null argument");
        if (p->shard_count != 1) throw std::invalid_argument("zk_prove on a sharded prover: use zk_prove_msm + zk_prove_finish");
        const DirectProof d{out, r32, s32, false};
        prove_msm(p, nullptr, wtns, nullptr, &d);
    });
}

int zk_prove_dev_submit(zk_prover *p, const void *d_wtns, const uint8_t *r32, const uint8_t *s32) {
    return guarded([&] {
        if (!p || !d_wtns) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(p->mtx);
        submit_locked(p, (const Fr *)d_wtns, nullptr, r32, s32);
    });
}

int zk_prove_submit(zk_prover *p, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32) {
    return guarded([&] {
        if (!p || !wtns) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(p->mtx);
        submit_locked(p, nullptr, wtns, r32, s32);
    });
}
int zk_prove_msm_collect(zk_prover *p, zk_msm_sums *partial) {
    return guarded([&] {
        if (!p || !partial) throw std::invalid_argument("null argument");
        collect_sums(p, partial);
    });
}

int zk_prove_collect(zk_prover *p, zk_proof *out) {
    return guarded([&] {
        if (!p || !out) throw std::invalid_argument("null argument");
        if (p->shard_count != 1) throw std::invalid_argument("zk_prove_collect on a sharded prover: use zk_prove_msm_collect + zk_prove_finish");
        const DirectProof d{out, nullptr, nullptr, true};
        collect_sums(p, nullptr, nullptr, &d);
    });
}

int zk_prove_batch_submit(zk_prover *p, const uint8_t *const *wtns, uint32_t count, const uint8_t *r32s, const uint8_t *s32s) {
    return guarded([&] {
        if (!p || !wtns || !count) throw std::invalid_argument("null argument");
        for (uint32_t k = 0; k < count; k++)
            if (!wtns[k]) throw std::invalid_argument("null witness pointer");
        if (p->shard_count != 1) throw std::invalid_argument("zk_prove_batch_submit on a sharded prover");
        if (count > p->batch) throw std::invalid_argument("more witnesses than the prover's opts.batch");
        std::lock_guard<std::mutex> lk(p->mtx);
        if (p->batch > 1) {
            const BatchIn bi{wtns, count, r32s, s32s};
            submit_locked(p, nullptr, nullptr, nullptr, nullptr, &bi);
        } else {
            submit_locked(p, nullptr, wtns[0], r32s, s32s);
        }
    });
}

int zk_prove_batch_collect(zk_prover *p, zk_proof *out, uint32_t count) {
    return guarded([&] {
        if (!p || !out || !count) throw std::invalid_argument("null argument");
        if (p->shard_count != 1) throw std::invalid_argument("zk_prove_batch_collect on a sharded prover");
        DirectProof d{out, nullptr, nullptr, true};
        d.count = count;
        if (p->batch == 1 && count != 1) throw std::invalid_argument("this prover was not created with opts.batch");
        collect_sums(p, nullptr, nullptr, &d);
    });
}
int zk_prover_timings(zk_prover *p, double *ms, uint32_t n) {
    return guarded([&] {
        if (!p || !ms) throw std::invalid_argument("null argument");
        if (!(p->flags & ZK_FLAG_TIMINGS)) throw std::invalid_argument("prover created without ZK_FLAG_TIMINGS");
        for (uint32_t i = 0; i < n && i < ZK_T_COUNT; i++) ms[i] = p->timings[i];
    });
}

}   // extern "C"
