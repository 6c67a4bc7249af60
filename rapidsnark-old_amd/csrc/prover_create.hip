// zk_prover_create / destroy / reserve / info: the MI355X replacement of Groth16::makeProver + the Prover constructor
// (reference src/groth16.cpp:9-46, src/groth16.hpp:57-95).  One-off work: CSR of the coefficient records, point tables
// (optionally window-precomputed), transform tables, streams; the per-proof workspace of slot 0.  No CPU fallback exists:
// every entry point fails if HIP does.
#include "prover_internal.hpp"

// A prover uses six HIP streams (two compute chains, two high-priority follow-up streams, the
// upload stream, the finishing stream) next to the application's own.  The HIP runtime multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with streams aliased onto one queue the
// witness upload of proof k+1 queued behind work of proof k and the proofs stopped overlapping —
// measured at 2^22 with host witnesses: 44.2 ms per proof with 4 queues, 36.9 with 8 (resident
// witnesses: 35.7); four provers on one GPU (24 streams) collapse to 350 ms per 2^16 proof with 8 queues
// and run at 2.2 ms with 16 or more, so the default asked for is 16 (no change at 2^22; the GPU has ~24
// hardware queue slots for ALL processes: beyond them the driver time-slices queues, so not more).  The variable is read when the HIP runtime initialises, so this only helps when
// the library is loaded before the process's first HIP call; hosts should export it themselves
// (INTEGRATION.md).  An explicit setting by the user is never overridden.
__attribute__((constructor)) static void zk_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

namespace zkp {

// Device and pinned-host workspace of one in-flight proof.
void alloc_slot(zk_prover *p, int i) {
    zk_prover::ProofSlot &q = p->slot[i];
    if (q.allocated) return;
    const uint64_t nv = p->sv.size();
    q.sort_w.alloc(nv, p->wbits, p->table_mode, p->batch);
    const MsmPlan pw = q.sort_w.plan, ph = p->sort_h.plan;
    const uint64_t tbw = q.sort_w.total_buckets(), tbh = p->sort_h.total_buckets();
    q.buckets_g1.alloc(3 * tbw + tbh);
    q.buckets_g2.alloc(tbw);
    q.scratch_g1.alloc(msm_reduce_scratch_points(3, pw) + msm_reduce_scratch_points(1, ph));
    const uint64_t ew = (uint64_t)pw.sets * msm_wsum_rc(pw), eh = (uint64_t)ph.sets * msm_wsum_rc(ph);     // window-sum records per MSM
    q.wsum_g1.alloc(3 * ew + eh);
    q.scratch_g2.alloc(msm_reduce_scratch_points(1, pw));
    q.wsum_g2.alloc(ew);
    const uint64_t slots = msm_accum_workspace_slots(q.sort_w.max_entries()), slots_h = msm_accum_workspace_slots(p->sort_h.max_entries());
    q.acc_stride = slots;
    q.acc_ws_g1_all.alloc(3 * slots + slots_h);
    q.acc_ws_g2.alloc(slots);
    q.acc_key_all.alloc(4 * slots + slots_h);
    q.acc_flag_all.alloc(4 * slots + slots_h);
    for (int m = 0; m < 5; m++) {          // A, B1, C at m*slots; H behind them; B2 last
        const uint64_t at = m < 3 ? m * slots : (m == 3 ? 3 * slots : 3 * slots + slots_h);
        if (m < 4) q.acc_ws_g1[m] = q.acc_ws_g1_all.p + at;
        q.acc_key[m] = q.acc_key_all.p + at;
        q.acc_flag[m] = q.acc_flag_all.p + at;
    }
    q.w1_bytes = (size_t)(3 * ew + eh) * sizeof(G1XYZZ);
    q.w2_bytes = (size_t)ew * sizeof(G2XYZZ);
    // (a call that ran out of memory half-way is repeated by the next submission: nothing below is made twice)
    if (!q.w1) HIP_TRY(hipHostMalloc((void **)&q.w1, q.w1_bytes, hipHostMallocDefault));
    if (!q.w2) HIP_TRY(hipHostMalloc((void **)&q.w2, q.w2_bytes, hipHostMallocDefault));
    for (auto &e : q.ev_l1)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t *e : {&q.ev_fork, &q.ev_join, &q.ev_sortw, &q.ev_main, &q.ev_done, &q.ev_chain})
        if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (auto &e : q.ev_tail)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : q.ev)
        if (!e) HIP_TRY(hipEventCreate(&e));
    q.have_events = true;
    q.allocated = true;
}

// the slot's HBM witness buffer and the events of its upload (host-witness proofs)
void ensure_witness_buffer(zk_prover *p, zk_prover::ProofSlot &q) {
    if (!q.wtns_dev.p) q.wtns_dev.alloc((uint64_t)p->nVars * p->batch);
    if (!q.ev_h2d) HIP_TRY(hipEventCreateWithFlags(&q.ev_h2d, hipEventDisableTiming));
    if (!q.ev_h2d_start) HIP_TRY(hipEventCreate(&q.ev_h2d_start));
}

// ZKHIP_VERBOSE=1: phase times of zk_prover_create on stderr (the one-shot CLI pays create on every run)
struct PhaseClock {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseClock() : on(getenv("ZKHIP_VERBOSE") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what, hipStream_t s) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[zkhip] create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

void prover_create(zk_prover **out, const zk_zkey_view *z, const zk_opts *o) {
    if (!out || !z) throw std::invalid_argument("null argument");
    PhaseClock clk;
    need_device_count();
    std::unique_ptr<zk_prover> p(new zk_prover());
    int dev = (o && o->device >= 0) ? o->device : -1;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    p->device = dev;
    DeviceGuard g(dev);
    p->flags = o ? o->flags : 0;
    p->shard_count = (o && o->shard_count > 1) ? o->shard_count : 1;
    p->shard_index = o ? o->shard_index : 0;
    p->batch = (o && o->batch > 1) ? o->batch : 1;
    if (p->batch > ZK_MAX_BATCH) throw std::invalid_argument("opts.batch > ZK_MAX_BATCH");
    if (p->flags & ZK_FLAG_PRECOMP_HALF) p->flags |= ZK_FLAG_PRECOMP;       // (a kind of pre-computation)
    if (p->batch > 1 && (!(p->flags & ZK_FLAG_PRECOMP) || (p->flags & ZK_FLAG_PRECOMP_HALF) || p->shard_count != 1 || (p->flags & ZK_FLAG_PARTITIONED_CHAIN)))
        throw std::invalid_argument("opts.batch needs ZK_FLAG_PRECOMP (a row per window) on an unsharded prover");
    if (p->shard_index >= p->shard_count) throw std::invalid_argument("shard_index >= shard_count");
    const uint32_t wbits = o ? o->window_bits : 0;

    p->nVars = z->nVars;
    p->nPublic = z->nPublic;
    p->domainSize = z->domainSize;
    p->nCoefs = z->nCoefs;
    if (z->nVars == 0 || z->nPublic + 1 > z->nVars) throw std::invalid_argument("invalid nVars/nPublic");
    p->logn = ilog2_exact(z->domainSize);
    // the coset shift needs a root of order 2*domainSize and BN254 Fr has 2-adicity 28 (the reference's
    // FFT<Fr>(2*domainSize), src/groth16.hpp:94, rejects larger domains the same way)
    if (p->logn > 27) throw std::invalid_argument("domainSize exceeds 2^27: the coset needs a root of order 2*domainSize and BN254 Fr has 2-adicity 28");
    // 32-bit index limits of the device data structures (all reachable on a 288 GB part)
    if (z->nCoefs >= (1ull << 32)) throw std::invalid_argument("nCoefs >= 2^32 is not supported (32-bit CSR positions)");
    const uint64_t n = z->domainSize, nV = z->nVars, nC = nV - z->nPublic - 1;
    // section size checks (the reference does none; an undersized section would be an OOB read)
    if (z->coefs_bytes && z->coefs_bytes < 4 + z->nCoefs * 44) throw std::invalid_argument("zkey section 4 too small");
    if (z->pointsA_bytes && z->pointsA_bytes < nV * 64) throw std::invalid_argument("zkey section 5 too small");
    if (z->pointsB1_bytes && z->pointsB1_bytes < nV * 64) throw std::invalid_argument("zkey section 6 too small");
    if (z->pointsB2_bytes && z->pointsB2_bytes < nV * 128) throw std::invalid_argument("zkey section 7 too small");
    if (z->pointsC_bytes && z->pointsC_bytes < nC * 64) throw std::invalid_argument("zkey section 8 too small");
    if (z->pointsH_bytes && z->pointsH_bytes < n * 64) throw std::invalid_argument("zkey section 9 too small");
    memcpy(p->vk_alpha1, z->vk_alpha1, 64);
    memcpy(p->vk_beta1, z->vk_beta1, 64);
    memcpy(p->vk_beta2, z->vk_beta2, 128);
    memcpy(p->vk_delta1, z->vk_delta1, 64);
    memcpy(p->vk_delta2, z->vk_delta2, 128);

    {
        // On a sharded prover the replicated SpMV + NTT chain on stream 1 is the critical path of a
        // rank (the MSM slices have shrunk, the chain has not): it gets the high priority of the
        // follow-up streams.  Measured per-rank time at 2^22 (one proof / two in flight): 2 shards
        // 24.5 -> 23.4 / 21.6 -> 21.6 ms, 8 shards 13.1 -> 12.6 / 10.4 -> 9.6 ms; unsharded it is
        // neutral to slightly negative (2^20: 13.4 -> 13.9 ms single) and stays off.  ZKHIP_S1_PRIO=0/1
        // overrides.
        const char *e = probe_env("ZKHIP_S1_PRIO");
        const bool s1_hi = e ? atoi(e) != 0 : p->shard_count >= 2;
        if (s1_hi && !getenv("ZKHIP_SERIAL")) {
            int lo_pr = 0, hi_pr = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_pr, &hi_pr));
            HIP_TRY(hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, hi_pr));
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        }
    }
    // The other five streams (a hardware queue each: 8-19 ms apiece, profiles/r03v_hip_init_probe.txt) are not used before
    // the first proof: a helper thread creates them while this one uploads the key (joined before create returns).
    struct SideStreams {
        std::thread th;
        std::exception_ptr err;
        ~SideStreams() {
            if (th.joinable()) th.join();
        }
        void finish() {
            if (th.joinable()) th.join();
            if (err) std::rethrow_exception(err);
        }
    } side;
    zk_prover *const pp = p.get();
    side.th = std::thread([pp, dev, &side] {
        try {
            zk_prover *const p = pp;
            HIP_TRY(hipSetDevice(dev));
            // ZKHIP_SERIAL=1 (profiling aid): one stream, so that rocprofv3 kernel durations are not
            // inflated by the other stream's kernels sharing the CUs.
            if (getenv("ZKHIP_SERIAL")) p->stream2 = p->stream;
            else HIP_TRY(hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking));
            {
                // Follow-up streams (highest priority) for the partial merges and bucket reductions of a SHARDED prover: on the
                // streams of their MSMs these small kernels queue behind the next level-1 launch and pile up after the last
                // one — with two proofs in flight the rank-0 share of 8 shards at 2^22 is 6.2-6.5 ms with them, 6.6 without
                // (round 1, before there were lanes: 2^20 unsharded 15.8 -> 13.5 ms).  An unsharded prover gets NONE since
                // round 3: its lanes do the same job (pipelined period equal with and without at 2^16 ... 2^22), and the
                // mere existence of the two high-priority queues costs a lone proof 2 % (2^22 synchronous 39.5 -> 38.8 ms,
                // 2^20 12.9 -> 12.0 ms, three-way same-box A/B).  ZKHIP_TAIL=0/2/5 overrides.
                // Two follow-up streams (the tails of stream 2's MSMs on one, of stream 1's on the other).  One per
                // MSM (ZKHIP_TAIL=5) was measured neutral at every size from 2^14 to 2^22 (tools/ab_tailstreams.sh)
                // — at most four kernels ever run concurrently in a proof's trace, whatever the number of streams —
                // and costs three more hardware queues.
                const char *e = getenv("ZKHIP_TAIL");
                int ntail = e ? atoi(e) : (p->shard_count > 1 ? 2 : 0);
                if (getenv("ZKHIP_SERIAL")) ntail = 0;
                if (ntail != 0 && ntail != 5) ntail = 2;
                p->tail_streams = ntail;
                if (ntail) {
                    int lo_pr = 0, hi_pr = 0;
                    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_pr, &hi_pr));
                    for (int i = 0; i < ntail; i++) HIP_TRY(hipStreamCreateWithPriority(&p->tail_pool[i], hipStreamNonBlocking, hi_pr));
                    for (int m = 0; m < 5; m++) p->tail[m] = ntail == 5 ? p->tail_pool[m] : p->tail_pool[(m == 2 || m == 3) ? 1 : 0];
                }
            }
            HIP_TRY(hipStreamCreateWithFlags(&p->stream_fin, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&p->stream_h2d, hipStreamNonBlocking));
        } catch (...) {
            side.err = std::current_exception();
        }
    });
    // window of the witness MSMs (sort(w), tables A / B1 / B2 / C); MSM H keeps `wbits` (its scalars are always full-size)
    p->wbits = (wbits == 0 && (p->flags & ZK_FLAG_SPARSE_WITNESS) && (p->flags & ZK_FLAG_PRECOMP) && p->logn > 18) ? 16u : wbits;
    hipStream_t s = p->stream;
    clk.lap("device + first stream", s);
    StreamUploader up(s);

    // --- this shard's contiguous slices of the witness indices and of the domain
    p->sv = shard_slice(nV, p->shard_index, p->shard_count);
    p->sh = shard_slice(n, p->shard_index, p->shard_count);
    p->part = (p->flags & ZK_FLAG_PARTITIONED_CHAIN) != 0 && p->shard_count > 1;
    if (p->part) {
        uint32_t lg = 0;
        while ((1u << lg) < p->shard_count) lg++;
        if ((1u << lg) != p->shard_count || lg > 3) throw std::invalid_argument("partitioned chain: shard_count must be 2, 4 or 8");
        if (p->logn < 2 * lg) throw std::invalid_argument("partitioned chain: domainSize must be at least shard_count^2");
        p->log_shards = lg;
    }
    p->nloc = p->part ? p->sh.size() : n;
    {
        // below ~2^20 a proof is bound by the latencies of its kernels, not by their work — and so are the
        // SHARDS of a larger one: A, B1 and C as one set of launches (level-1, merges, ONE reduction over three bucket
        // sets).  Same box, probes build (profiles/r04e_shard8_experiments.txt, r04f_batch_abc_experiments.txt): rank-0 share of
        // 8 shards of 2^22 7.22 -> 6.48 ms one at a time / 6.39 -> 6.10 two in flight, 4 shards 13.8 -> 11.5 / 10.9 -> 10.3,
        // 2 shards 20.7 -> 20.5 / 19.5 -> 18.5, 8 shards of 2^24 20.7 -> 20.3 / 19.4 -> 18.6; unsharded 2^19 5.06 -> 4.86 ms.
        // And the LARGE unsharded circuits too (profiles/r04h_ab_batch_abc_unsharded.txt, r04i_ab_batch_abc_sync_2p22.txt, three
        // alternations each): 2^22 period with resident witnesses 32.9 -> 32.4 ms (-1.4 %), one synchronous zk_prove 37.6 ->
        // 37.0; 2^24 131.9 -> 130.1; tables as in the zkey 38.7 -> 37.6; circuit-shaped key with a realistic witness one at a
        // time 13.9 -> 12.7.  Only 2^20 and 2^21 unsharded measured neutral on the period and 0.3-0.5 ms WORSE for a lone proof
        // (three level-1 launches of 1-2 ms interleave with the other stream's work, one of 3-6 ms does not): off there.
        const char *e = getenv("ZKHIP_BATCH_ABC");
        // Round 5 (busy proofs run fewer, longer lanes; c = 20 at 2^21): 2^21 now gains 1.8 % on the period with a lone proof equal
        // (profiles/r05zp_batch_abc_mid_sizes.txt, three alternations), 2^20 is still neutral: off at 2^20 only.
        const bool mid_size_unsharded = p->shard_count == 1 && p->sv.size() >= (1u << 20) && p->sv.size() < (1u << 21);
        p->batch_abc = e ? atoi(e) != 0 : !mid_size_unsharded;
    }
    HIP_TRY(hipEventCreateWithFlags(&p->ev_ext_in, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_ext_out, hipEventDisableTiming));

    // --- CSR (src/groth16.cpp:38: records start 4 bytes into section 4), built on the device from
    // the raw records (the host pass over 4n random rows was the largest single part of create);
    // a partitioned prover keeps the rows of its own block only
    {
        const uint64_t nnz = z->nCoefs;
        const uint32_t rows = 2 * (uint32_t)p->nloc;
        DevBuf<uint8_t> raw;
        DevBuf<uint32_t> cursor, err;
        raw.alloc(nnz ? nnz * 44 : 4);
        cursor.alloc(rows);
        err.alloc(1);
        p->csr_rowptr.alloc((size_t)rows + 1 + msm_scan_extra_words(rows));
        p->csr_col.alloc(nnz ? nnz : 1);
        p->csr_val.alloc(nnz ? nnz : 1);
        if (nnz) up.copy(raw.p, (const uint8_t *)z->coefs + 4, nnz * 44);
        clk.lap("coefficient records upload", s);
        launch_csr_build(p->csr_rowptr.p, p->csr_col.p, p->csr_val.p, cursor.p, err.p, raw.p, nnz, z->domainSize, z->nVars,
                         p->part ? (uint32_t)p->sh.lo : 0u, p->part ? (uint32_t)p->sh.hi : z->domainSize, s);
        launch_fr_to_internal(p->csr_val.p, nnz, 2, s);      // value*2^512 -> value*2^522 (see k_spmv_abc); unused tail entries are zero
        uint32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, err.p, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (bad) throw std::invalid_argument("zkey coefficient record out of range");
        clk.lap("CSR build (device)", s);
    }

    // --- twiddles.  The proof path runs the register-butterfly pipeline of nttpair.hip on this prover's block (its own
    // tables); the radix-2 tables of ntt.hip are only needed by the cross-GPU stages of a partitioned chain (full-size
    // twiddles) and where the pipeline does not apply (blocks of fewer than 8 elements)
    {
        const uint32_t local_logn = p->logn - p->log_shards_chain();
        const bool pair = ntt_pair_supported(local_logn) && !probe_env("ZKHIP_NTT_RADIX2");
        if (pair) p->pair.build(p->logn, local_logn, p->part ? p->shard_index : 0u, s);
        if (!pair || p->part) {
            p->tw_fwd.alloc(n > 1 ? n / 2 : 1);
            p->tw_inv.alloc(n > 1 ? n / 2 : 1);
            if (!pair) p->tw_coset.alloc(n);
            p->tw_ninv.alloc(1);
            launch_ntt_build_tables(p->tw_fwd.p, p->tw_inv.p, p->tw_coset.p, p->tw_ninv.p, p->logn, s);
        }
    }
    clk.lap("twiddle tables", s);

    // --- point tables: this shard's contiguous slices
    const uint64_t nv = p->sv.size(), nh = p->sh.size();
    p->precomp = (p->flags & ZK_FLAG_PRECOMP) != 0;
    p->table_mode = (p->flags & ZK_FLAG_PRECOMP_HALF) ? 2u : p->precomp ? 1u : 0u;
    p->sort_h.alloc(nh, wbits, p->table_mode, p->batch);
    alloc_slot(p.get(), 0);
    // with window pre-computation a table holds W rows: row j = 2^(c*j) * P (msm.hip) — or, with ZK_FLAG_PRECOMP_HALF, the
    // ceil(W/2) rows of the even windows
    const uint64_t rows_w = msm_table_rows(p->slot[0].sort_w.plan), rows_h = msm_table_rows(p->sort_h.plan);
    p->ptsA.alloc((nv ? nv : 1) * rows_w);
    p->ptsB1.alloc((nv ? nv : 1) * rows_w);
    p->ptsB2.alloc((nv ? nv : 1) * rows_w);
    p->ptsH.alloc((nh ? nh : 1) * rows_h);
    clk.lap("workspace allocation", s);
    up.copy(p->ptsA.p, (const uint8_t *)z->pointsA + p->sv.lo * 64, nv * 64);
    up.copy(p->ptsB1.p, (const uint8_t *)z->pointsB1 + p->sv.lo * 64, nv * 64);
    up.copy(p->ptsB2.p, (const uint8_t *)z->pointsB2 + p->sv.lo * 128, nv * 128);
    up.copy(p->ptsH.p, (const uint8_t *)z->pointsH + p->sh.lo * 64, nh * 64);
    // C: witness index i (global) uses pointsC[i - nPublic - 1] for i > nPublic (src/groth16.cpp:204)
    {
        uint64_t first = z->nPublic + 1;                 // first global witness index with a C point
        uint64_t lo = p->sv.lo > first ? p->sv.lo : first;
        uint64_t hi = p->sv.hi > lo ? p->sv.hi : lo;
        uint64_t cnt = hi - lo;
        uint32_t skip = (uint32_t)(lo - p->sv.lo);       // leading witness rows of this shard without a C point
        if (p->precomp) {
            // same row indexing as A/B1 (entries address row j*nv + i): pad the public rows with infinity
            p->ptsC.alloc((nv ? nv : 1) * rows_w);
            HIP_TRY(hipMemsetAsync(p->ptsC.p, 0, (size_t)(nv ? nv : 1) * 64, s));
            if (cnt) up.copy(p->ptsC.p + skip, (const uint8_t *)z->pointsC + (lo - first) * 64, cnt * 64);
            p->c_idx_min = 0;
            launch_fq_to_internal((Fq *)p->ptsC.p, nv * 2, s);
        } else {
            p->c_idx_min = skip;
            p->ptsC.alloc(cnt ? cnt : 1);
            up.copy(p->ptsC.p, (const uint8_t *)z->pointsC + (lo - first) * 64, cnt * 64);
            launch_fq_to_internal((Fq *)p->ptsC.p, cnt * 2, s);
        }
    }
    // MSM kernels work in the 2^261 Montgomery form (field29.hpp): convert the tables once
    launch_fq_to_internal((Fq *)p->ptsA.p, nv * 2, s);
    launch_fq_to_internal((Fq *)p->ptsB1.p, nv * 2, s);
    launch_fq_to_internal((Fq *)p->ptsB2.p, nv * 4, s);
    launch_fq_to_internal((Fq *)p->ptsH.p, nh * 2, s);
    clk.lap("point tables upload+convert", s);
    if (p->precomp) {
        // one scratch area for the doubling walks, reused table after table (freed on return)
        const MsmPlan plan_w = p->slot[0].sort_w.plan;
        const uint64_t tw = (uint64_t)(rows_w - 1) * (nv ? nv : 1), th = (uint64_t)(rows_h - 1) * (nh ? nh : 1);
        const uint64_t tmax = tw > th ? tw : th;
        DevBuf<G2XYZZ> tmp;
        DevBuf<Fq2> pref;
        tmp.alloc(tmax ? tmax : 1);
        pref.alloc(tmax ? tmax : 1);
        launch_msm_precomp_g1(p->ptsA.p, (G1XYZZ *)tmp.p, (Fq *)pref.p, nv, plan_w, s);
        launch_msm_precomp_g1(p->ptsB1.p, (G1XYZZ *)tmp.p, (Fq *)pref.p, nv, plan_w, s);
        launch_msm_precomp_g1(p->ptsC.p, (G1XYZZ *)tmp.p, (Fq *)pref.p, nv, plan_w, s);
        launch_msm_precomp_g1(p->ptsH.p, (G1XYZZ *)tmp.p, (Fq *)pref.p, nh, p->sort_h.plan, s);
        launch_msm_precomp_g2(p->ptsB2.p, tmp.p, pref.p, nv, plan_w, s);
        HIP_TRY(hipStreamSynchronize(s));
    }

    // --- workspace (slot 0 was allocated above; slot 1 appears with the first overlapped submit)
    p->abc.alloc(3 * p->nloc * p->batch);
    p->h.alloc(p->nloc * p->batch);
    if (p->part) p->xb.alloc(3 * p->nloc);
    p->abc_use = p->abc.p;
    p->xb_use = p->xb.p;
    {
        // lanes for small circuits (see zk_prover::LaneExtra).  ZKHIP_LANES=1..8 overrides.  Eight up to 2^16 with one witness
        // per submission: a proof of that size is two chains of ~28 launches whose kernels fill a fraction of the chip each, and
        // what bounds it is how many chains run side by side (profiles/r04aj_lanes.txt: 0.72 -> 0.64 ms at 2^14, 1.16 -> 1.08 at
        // 2^16, nothing from 2^18 on; batched submissions are SLOWER with eight: 0.39 -> 0.62 ms at 2^14 x 4).  With the wave priorities
        // in place (common.hpp) 2^17 gains too: 1.60 -> 1.53 ms; 2^18 equal, 2^19 +1.5 % (profiles/r04bm_lanes_with_priorities.txt).
        const char *e = getenv("ZKHIP_LANES");
        int lanes = e ? atoi(e) : (p->domainSize <= (1u << 17) && p->batch == 1 ? 8 : p->domainSize <= (1u << 22) ? 4 : 1);
        if (lanes < 1) lanes = 1;
        if (lanes > zk_prover::MAX_LANES) lanes = zk_prover::MAX_LANES;
        if (p->part || getenv("ZKHIP_SERIAL")) lanes = 1;
        for (int l = 1; l < lanes; l++) {
            auto x = std::make_unique<zk_prover::LaneExtra>();
            const char *ls = getenv("ZKHIP_LANE_STREAMS");
            x->one_stream = ls && atoi(ls) == 1;
            x->n_abc = 3 * p->nloc * p->batch;
            x->n_h = p->nloc * p->batch;
            x->nh_sort = nh;
            x->wbits = wbits;
            x->precomp = p->table_mode;
            x->batch = p->batch;
            p->extra[l - 1] = std::move(x);
        }
        p->lanes = lanes;
        const char *ge = getenv("ZKHIP_GRAPH");
        p->use_graph = ge && atoi(ge) != 0 && !p->part && p->batch == 1 && !(p->flags & ZK_FLAG_TIMINGS) && !getenv("ZKHIP_SERIAL");
    }
    HIP_TRY(hipStreamSynchronize(s));   // host image may be released after return
    side.finish();
    clk.lap(p->precomp ? "window pre-computation" : "finish", s);
    *out = p.release();
}

}   // namespace zkp

extern "C" {

int zk_device_count(int *count) {
    return guarded([&] {
        int n = 0;
        HIP_TRY(hipGetDeviceCount(&n));
        *count = n;
    });
}

int zk_prover_create(zk_prover **out, const zk_zkey_view *zkey, const zk_opts *opts) {
    return guarded([&] { prover_create(out, zkey, opts); });
}

void zk_prover_destroy(zk_prover *p) {
    if (!p) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->device);
    delete p;
    if (prev >= 0) (void)hipSetDevice(prev);
}

int zk_host_alloc(void **out, size_t bytes) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("null argument");
        need_device_count();
        HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    });
}

void zk_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int zk_prover_reserve(zk_prover *p, uint32_t in_flight, uint32_t host_witnesses) {
    return guarded([&] {
        if (!p) throw std::invalid_argument("null argument");
        if (in_flight > ZK_MAX_IN_FLIGHT) throw std::invalid_argument("in_flight > ZK_MAX_IN_FLIGHT");
        std::lock_guard<std::mutex> lk(p->mtx);
        DeviceGuard g(p->device);
        // A pipeline of depth d keeps d proofs in flight and submits the next one as soon as the oldest is collected: it walks
        // d + 1 slots of the ring at most when a collect and a submit overlap (slot i runs on lane i % lanes); one proof at a
        // time lives in slot 0.  The ring is cut to that length, so a depth-3 pipeline of a 2^24 circuit reserves four slots and
        // not eight (each costs GiBs there: reserving all of them turned a pipeline that fits into a start-up out-of-memory).
        const int nslots = in_flight >= 2 ? (int)std::min<uint32_t>(ZK_MAX_IN_FLIGHT, in_flight + 1) : 1;
        if (p->in_flight || p->phase_open >= 0) throw std::invalid_argument("proofs in flight");
        p->ring = nslots == 1 ? ZK_MAX_IN_FLIGHT : (uint32_t)nslots;
        p->next_submit = p->next_collect = 0;
        // What an earlier, deeper reservation that ran out of memory left behind (slots and lanes beyond this ring) goes back
        // first: the retry with a shallower pipeline (Groth16::makeProver's fallback chain) must see the memory a fresh
        // prover would see.  Nothing is in flight (checked above), so nothing is using them.
        HIP_TRY(hipDeviceSynchronize());
        for (int i = nslots; i < (int)ZK_MAX_IN_FLIGHT; i++)
            p->slot[i].release_memory();
        for (int lane = nslots < 1 ? 1 : nslots; lane < p->lanes; lane++)
            if (p->extra[lane - 1]) p->extra[lane - 1]->release_memory();
        // Every slot of the ring is allocated NOW (device workspace, and for host witnesses the HBM witness buffer and its
        // pinned staging copy), so that out-of-memory is a start-up error and the first `depth` proofs do not pay for it.
        for (int i = 0; i < nslots; i++) {
            alloc_slot(p, i);
            if (host_witnesses) {
                zk_prover::ProofSlot &q = p->slot[i];
                ensure_witness_buffer(p, q);
                if (!q.wtns_pin) HIP_TRY(hipHostMalloc((void **)&q.wtns_pin, (size_t)p->nVars * 32 * p->batch, hipHostMallocDefault));
            }
        }
        for (int lane = 1; lane < p->lanes && lane < nslots; lane++) p->extra[lane - 1]->ensure();
    });
}

int zk_prover_info(zk_prover *p, zk_prover_plan *plan) {
    return guarded([&] {
        if (!p || !plan) throw std::invalid_argument("null argument");
        if (plan->size < 8 || plan->size > 4096) throw std::invalid_argument("zk_prover_plan.size not set");
        zk_prover_plan o;
        memset(&o, 0, sizeof o);
        const MsmPlan ph = p->sort_h.plan, pw = p->slot[0].sort_w.plan;
        o.window_bits_h = ph.c; o.windows_h = ph.W;
        o.window_bits_w = pw.c; o.windows_w = pw.W;
        o.precomputed_tables = p->table_mode;
        o.table_rows_h = msm_table_rows(ph); o.table_rows_w = msm_table_rows(pw);
        o.bucket_sets_h = ph.sets; o.bucket_sets_w = pw.sets;
        o.msm_a_b1_c_one_launch = p->batch_abc ? 1u : 0u;
        o.lanes = (uint32_t)p->lanes;
        o.follow_up_streams = (uint32_t)p->tail_streams;
        o.max_in_flight = ZK_MAX_IN_FLIGHT;
        // Small circuits are bound by the serial latency of their ~60 kernels: the maximum.  Up to 2^22 (four lanes) six hide
        // the upload of a host witness completely (2^22: four / five / six in flight 33.5 / 32.6 / 32.5 ms).  Above, and on a
        // shard, two saturate the chip and a third hides the upload (each slot costs GiBs there).
        const bool small = p->logn < 19, mid = p->logn <= 22 && p->shard_count == 1;
        o.depth_host_witness = small ? ZK_MAX_IN_FLIGHT : mid ? 6u : 3u;
        o.depth_resident_witness = small ? ZK_MAX_IN_FLIGHT : mid ? 6u : 2u;
        o.batch = p->batch;
        o.shard_index = p->shard_index; o.shard_count = p->shard_count; o.chain_partitioned = p->part ? 1u : 0u;
        {
            DeviceGuard g(p->device);
            size_t fr = 0, tot = 0;
            HIP_TRY(hipMemGetInfo(&fr, &tot));
            o.device_bytes_in_use = tot - fr; o.device_bytes_total = tot;
        }
        o.kernel_launches_last_proof = p->launches_last_proof;
        const uint32_t nb = plan->size < sizeof o ? plan->size : (uint32_t)sizeof o;
        o.size = nb;
        memcpy(plan, &o, nb);
    });
}

}   // extern "C"
