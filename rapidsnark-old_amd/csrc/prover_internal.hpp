// Internal header of libzkhip's host side (not part of the C-ABI: include/zkhip.h is).  The prover object, the small
// RAII helpers around HIP memory and the functions the four translation units call in each other:
//   prover_create.hip    zk_prover_create / destroy / reserve / info: one-off work of Groth16::makeProver (src/groth16.cpp:9-46)
//   prover_pipeline.hip  one proof as enqueued phases, submit / collect, the synchronous entry points (src/groth16.cpp:48-254)
//   prover_multi.hip     one proof on several GPUs: zk_multi_prover (one process) and zk_shard_* (one process per GPU)
//   operators.hip        operator-level entry points (zk_fr_ntt, zk_msm_g1, ...) and the synthetic-table helpers
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <sys/random.h>
#include <string>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <stdexcept>
#include <exception>
#include <functional>

#include "../../include/zkhip.h"
#include "common.hpp"
#include "hipcheck.hpp"
#include "kernels.hpp"
#include "tail_pool.hpp"

using namespace zk;

namespace zkp {

#define HIP_TRY(expr) ZK_HIP(expr)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) HIP_TRY(hipMalloc((void **)&p, count * sizeof(T)));
    }
    void upload(const void *src, size_t count, hipStream_t s) {
        if (count) HIP_TRY(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
};

// Host image -> HBM for the big zkey sections (src/binfile_utils.cpp:28-33 copies the whole file into a
// malloc'ed image first; here the image is the caller's, normally a read-only mmap of the .zkey: pageable
// and possibly not yet in the page cache).  Two pinned staging chunks: while chunk k's DMA runs, four host
// threads pull chunk k+1 out of the mapping (page faults / disk reads happen there, off the DMA's path).
// A source that is already page-locked is copied from directly.
struct StreamUploader {
    static constexpr size_t CHUNK = (size_t)64 << 20;
    uint8_t *pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t s;
    int k = 0;
    explicit StreamUploader(hipStream_t s_) : s(s_) {}
    ~StreamUploader() {
        for (int i = 0; i < 2; i++) {
            if (done[i]) {
                (void)hipEventSynchronize(done[i]);
                (void)hipEventDestroy(done[i]);
            }
            if (pin[i]) (void)hipHostFree(pin[i]);
        }
    }
    void copy(void *dst, const void *src, size_t bytes) {
        if (!bytes) return;
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (pinned || bytes < ((size_t)4 << 20)) {
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
            return;
        }
        for (size_t off = 0; off < bytes; off += CHUNK, k ^= 1) {
            const size_t len = bytes - off < CHUNK ? bytes - off : CHUNK;
            if (!pin[k]) {
                HIP_TRY(hipHostMalloc((void **)&pin[k], CHUNK, hipHostMallocDefault));
                HIP_TRY(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            } else {
                HIP_TRY(hipEventSynchronize(done[k]));          // the DMA that last read this chunk
            }
            const uint8_t *from = (const uint8_t *)src + off;
            uint8_t *to = pin[k];
            const size_t nt = 4, per = (len / nt + 4095) & ~(size_t)4095;
            std::vector<std::thread> th;
            for (size_t t = 1; t < nt; t++) {
                const size_t lo = t * per, hi = lo + per < len ? lo + per : len;
                if (lo < hi) th.emplace_back([=] { memcpy(to + lo, from + lo, hi - lo); });
            }
            memcpy(to, from, per < len ? per : len);
            for (auto &t : th) t.join();
            HIP_TRY(hipMemcpyAsync((uint8_t *)dst + off, to, len, hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(done[k], s));
        }
    }
};

struct Slice {
    uint64_t lo, hi;
    uint64_t size() const { return hi - lo; }
};
inline Slice shard_slice(uint64_t n, uint32_t idx, uint32_t cnt) {
    uint64_t per = (n + cnt - 1) / cnt;
    uint64_t lo = per * idx, hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    return Slice{lo, hi};
}

// Scalar-vector sort workspace (one per scalar set: witness, h)
struct SortBufs {
    MsmPlan plan;
    uint64_t n = 0;
    DevBuf<uint16_t> lo;
    DevBuf<uint32_t> counts, starts, offsets, entries, codes, val, bin_counts, bin_starts;
    uint32_t total_buckets() const { return plan.sets * plan.nbuckets; }
    uint64_t max_entries() const { return (n ? n : 1) * plan.W; }
    // batch > 1: `batch` scalar vectors of n_ scalars each, sorted together into one bucket set per vector
    // precomp: 0 = tables as in the zkey, 1 = a table row per window, 2 = a row per second window (MsmPlan::precomp)
    void alloc(uint64_t n_, uint32_t window_bits, uint32_t precomp = 0, uint32_t batch = 1) {
        plan = make_msm_plan(n_ ? n_ : 1, window_bits, precomp, batch);
        n = n_ * (batch > 1 ? batch : 1);
        // sort entries are 32-bit (bit 31 = digit sign): positions n*W and, with window-precomputed
        // tables, table rows j*n + i must stay below 2^32 / 2^31
        if ((n ? n : 1) * plan.W >= (1ull << 32)) throw std::invalid_argument("MSM too large: n * windows >= 2^32 sort entries");
        if ((n ? n : 1) * msm_table_rows(plan) >= (1ull << 31)) throw std::invalid_argument("MSM too large: table rows >= 2^31");
        MsmSortSizes z = msm_sort_sizes(n, plan);
        lo.alloc(z.lo_u16);
        counts.alloc(z.counts_u32);
        starts.alloc(z.starts_u32);
        offsets.alloc(z.offsets_u32);
        entries.alloc(z.entries_u32);
        codes.alloc(z.codes_u32);
        val.alloc(z.val_u32);
        bin_counts.alloc(z.bin_counts_u32);
        bin_starts.alloc(z.bin_starts_u32);
    }
    void release() {
        lo.release(); counts.release(); starts.release(); offsets.release(); entries.release();
        codes.release(); val.release(); bin_counts.release(); bin_starts.release();
        n = 0; ran = false;
    }
    bool ran = false;
    void run(const Fr *scalars, hipStream_t s) {
        // ZKHIP_PROBE_SKIP_SORT=1 (-DZK_PROBES builds only; wrong sums): the sort runs once per buffer set and its result is reused —
        // what a proof costs if digits + counting sort were free (profiles/NEGATIVE_RESULTS.md item 22)
        static const bool skip = probe_env("ZKHIP_PROBE_SKIP_SORT") != nullptr;
        if (skip && ran) return;
        ran = true;
        MsmSortBufs b{offsets.p, entries.p, counts.p, starts.p, codes.p, val.p, bin_counts.p, bin_starts.p, lo.p};
        launch_msm_sort(b, scalars, n, plan, s);
    }
};


}   // namespace zkp
using namespace zkp;

// pageable witness -> pinned staging, run as a host function on the upload stream
struct StageJob {
    uint8_t *dst;
    const uint8_t *src;
    size_t bytes;
};

struct zk_prover {
    int device = 0;
    uint32_t flags = 0;
    uint32_t nVars = 0, nPublic = 0, domainSize = 0, logn = 0;
    uint64_t nCoefs = 0;
    uint32_t shard_index = 0, shard_count = 1;
    uint8_t vk_alpha1[64], vk_beta1[64], vk_beta2[128], vk_delta1[64], vk_delta2[128];
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: witness-only MSM chain (A,B1,C,B2)
    // mtx: submission + slot bookkeeping.  cmtx: serialises collectors; a collect holds mtx only to look the
    // slot up and to retire it, NOT while it waits for the GPU and runs the host tail (0.3 ms: window sums +
    // final assembly) — a second thread collecting while the first submits keeps both off each other's path.
    // sync_mtx: one synchronous zk_prove* call at a time.
    std::mutex mtx, cmtx, sync_mtx;

    // resident data
    DevBuf<uint32_t> csr_rowptr, csr_col;
    DevBuf<Fr> csr_val;
    DevBuf<TwEntry> tw_fwd, tw_inv;
    DevBuf<Fr> tw_coset, tw_ninv;
    NttPair pair;               // nttpair.hip: tables of the coset-evaluation pipeline for this prover's block (pair.L == 0: not used)
    Slice sv, sh;              // this shard's slice of witness indices / domain indices
    uint32_t c_idx_min = 0;    // C-MSM: local witness index >= c_idx_min maps to pointsC[idx - c_idx_min]
    bool precomp = false;      // window-precomputed tables (ZK_FLAG_PRECOMP): tables hold W rows of n points ...
    uint32_t table_mode = 0;   // ... (1), or ceil(W/2) rows with ZK_FLAG_PRECOMP_HALF (2); 0 = tables as in the zkey.  What SortBufs::alloc takes.
    DevBuf<G1Affine> ptsA, ptsB1, ptsC, ptsH;
    DevBuf<G2Affine> ptsB2;

    // per-proof workspace used on `stream` only (in-order across consecutive proofs)
    DevBuf<Fr> abc, h;         // abc = a|b|c back to back
    SortBufs sort_h;
    // Everything a proof's witness-side streams and its asynchronous follow-up kernels touch lives
    // in a ProofSlot; two slots let the front of proof k+1 (sort, SpMV, NTT: LDS/latency-bound)
    // overlap the tail of proof k (merges, reductions, D2H, host Horner + assembly).
    struct ProofSlot {
        bool allocated = false, busy = false;
        bool use_tails = true;                // this proof's merges / reductions on the follow-up streams (decided at submit)
        uint32_t l1_chunk_min = 0, l1_chunk_max = 0;      // entries per level-1 lane: at least / per round at most (0: msm.hip's defaults), decided at submit
        SortBufs sort_w;
        DevBuf<G1Acc> buckets_g1;    // A | B1 | C | H   (A,B1,C use sort_w's plan; H uses sort_h's)
        DevBuf<G2Acc> buckets_g2;
        // accumulation workspaces, one per MSM: 0 = A, 1 = B1, 2 = C, 3 = H (G1), 4 = B2 (G2)
        DevBuf<G1Acc> scratch_g1, acc_ws_g1_all;       // accumulation workspaces of MSM A | B1 | C (equal strides: batched launches) | H
        G1Acc *acc_ws_g1[4] = {nullptr, nullptr, nullptr, nullptr};
        DevBuf<uint32_t> acc_key_all, acc_flag_all;    // A | B1 | C | H | B2
        uint32_t *acc_key[5] = {nullptr}, *acc_flag[5] = {nullptr};
        uint64_t acc_stride = 0;
        DevBuf<G2Acc> scratch_g2, acc_ws_g2;
        DevBuf<G1XYZZ> wsum_g1;
        DevBuf<G2XYZZ> wsum_g2;
        hipEvent_t ev_l1[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_sortw = nullptr, ev_main = nullptr, ev_done = nullptr, ev_chain = nullptr;
        bool zeroed = false;                  // this proof's bucket arrays were cleared at submit, beside the witness upload (phase_front)
        bool defer_w = false;                 // lone proof: the witness MSMs are enqueued behind the transform chain (phase_local), see phase_front
        hipEvent_t ev_tail[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipEvent_t ev[20] = {};  // 0-6 stage marks; 8/9, 13/14, 15/16, 17/18: G1 level-1 kernels of MSM A, B1, C, H; 10/11: G2; 12: upload
        bool have_events = false;
        uint8_t *w1 = nullptr, *w2 = nullptr;      // pinned host copies of the window sums
        size_t w1_bytes = 0, w2_bytes = 0;
        // host-witness proofs (zk_prove / zk_prove_submit): the witness of THIS proof in HBM, and the
        // pinned staging copy a pageable caller buffer goes through.  One per slot, so that proof
        // k+1's upload runs (on its own stream) while proof k is still computing.
        DevBuf<Fr> wtns_dev;              // batch x nVars
        uint8_t *wtns_pin = nullptr;
        uint8_t *pin_ring = nullptr;          // two upload pieces of pinned memory: the staging of a lone proof on a slot that has no wtns_pin yet
        StageJob stage[ZK_MAX_BATCH];
        StageJob stage_chunk[16];        // a large pageable witness is staged and uploaded in pieces on two streams
        hipEvent_t ev_h2d_b = nullptr;
        hipEvent_t ev_h2d = nullptr, ev_h2d_start = nullptr;
        uint8_t r32[ZK_MAX_BATCH][32], s32[ZK_MAX_BATCH][32];
        bool have_r = false, have_s = false;
        bool host_witness = false;
        uint32_t count = 1;               // proofs this submission carries (<= the prover's batch)
        // small circuits: the ~60 launches of a proof captured once as a HIP graph (per slot: every pointer in it
        // is the slot's or the lane's) and replayed; valid for the witness address it was captured with
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        const Fr *graph_wtns = nullptr;
        hipEvent_t ev_gdone = nullptr;      // recorded behind the graph launch: what a collect waits for
        bool via_graph = false;
        // Give the slot's device and pinned memory back (events stay).  Only for a slot no proof is using: zk_prover_reserve
        // does this to the slots beyond a ring that a failed, deeper reservation left allocated.
        void release_memory() {
            allocated = false;
            sort_w.release();
            buckets_g1.release(); buckets_g2.release(); scratch_g1.release(); acc_ws_g1_all.release();
            acc_key_all.release(); acc_flag_all.release(); scratch_g2.release(); acc_ws_g2.release();
            wsum_g1.release(); wsum_g2.release(); wtns_dev.release();
            if (w1) { (void)hipHostFree(w1); w1 = nullptr; }
            if (w2) { (void)hipHostFree(w2); w2 = nullptr; }
            if (wtns_pin) { (void)hipHostFree(wtns_pin); wtns_pin = nullptr; }
            if (pin_ring) { (void)hipHostFree(pin_ring); pin_ring = nullptr; }
            if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
            if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
            graph_wtns = nullptr;
        }
        ~ProofSlot() {
            if (gexec) (void)hipGraphExecDestroy(gexec);
            if (graph) (void)hipGraphDestroy(graph);
            if (ev_gdone) (void)hipEventDestroy(ev_gdone);
            for (auto &e : ev_l1) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : {ev_fork, ev_join, ev_sortw, ev_main, ev_done, ev_chain}) if (e) (void)hipEventDestroy(e);
            for (auto &e : ev_tail) if (e) (void)hipEventDestroy(e);
            for (auto &e : ev) if (e) (void)hipEventDestroy(e);
            if (w1) (void)hipHostFree(w1);
            if (w2) (void)hipHostFree(w2);
            if (wtns_pin) (void)hipHostFree(wtns_pin);
            if (pin_ring) (void)hipHostFree(pin_ring);
            if (ev_h2d) (void)hipEventDestroy(ev_h2d);
            if (ev_h2d_b) (void)hipEventDestroy(ev_h2d_b);
            if (ev_h2d_start) (void)hipEventDestroy(ev_h2d_start);
        }
    };
    ProofSlot slot[ZK_MAX_IN_FLIGHT];
    uint32_t next_submit = 0, next_collect = 0, in_flight = 0;
    uint32_t ring = ZK_MAX_IN_FLIGHT;   // slots the submissions cycle through (zk_prover_reserve cuts it to the pipeline's depth + 1)
    uint32_t batch = 1;         // opts.batch: witnesses proved by one submission (one set of kernel launches)
    // Small circuits: a proof is ~60 launches of kernels that each fill a tenth of the chip and wait on a
    // serial chain of point additions, so throughput comes from running SEVERAL PROOFS' kernels at once.
    // Consecutive proofs on the same streams cannot (stream order); `lanes` independent sets of
    // {stream, stream2, a|b|c, h, sort(h) buffers} can: slot i uses lane i % lanes.  Lane 0 is the prover's
    // own streams and buffers above; the extra lanes run their follow-up kernels and the final copies on
    // their own two streams (hardware queues are few: csrc/prover_create.hip, GPU_MAX_HW_QUEUES).
    struct LaneExtra {
        hipStream_t stream = nullptr, stream2 = nullptr;
        DevBuf<Fr> abc, h;
        SortBufs sort_h;
        // the streams (a hardware queue each: ~8 ms to create) and the buffers (~1 GiB per lane at 2^22) appear the first
        // time a proof runs on the lane, like the proof slots: the one-shot CLI and the shards of a sharded proof never
        // use more than lane 0
        uint64_t n_abc = 0, n_h = 0, nh_sort = 0;
        uint32_t wbits = 0, batch = 1;
        uint32_t precomp = 0;       // the prover's table_mode
        bool one_stream = false;
        bool ready = false;         // set at the END of ensure(), like ProofSlot::allocated: a call that ran out of memory half-way
                                    // (six proofs in flight at 2^22 next to nearly full tables) is repeated by the next proof on the
                                    // lane instead of leaving h / sort_h null behind a non-null abc
        void ensure() {
            if (ready) return;
            if (!stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            if (!stream2) {
                if (one_stream) stream2 = stream;
                else HIP_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
            }
            if (!abc.p) abc.alloc(n_abc);
            if (!h.p) h.alloc(n_h);
            sort_h.alloc(nh_sort, wbits, precomp, batch);      // (releases what an interrupted call left, then allocates all nine buffers)
            ready = true;
        }
        void release_memory() {      // buffers only: the streams (hardware queues) are cheap to keep and slow to make
            ready = false;
            abc.release(); h.release(); sort_h.release();
        }
        ~LaneExtra() {
            if (stream2 && stream2 != stream) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
            if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        }
    };
    static constexpr int MAX_LANES = 8;
    std::unique_ptr<LaneExtra> extra[MAX_LANES - 1];
    int lanes = 1;
    uint32_t wbits = 0;
    // ---- chain partitioned across the shards (ZK_FLAG_PARTITIONED_CHAIN; shard_count = 2^log_shards):
    // this prover computes rows [sh.lo, sh.hi) of a, b, c only, runs the local stages of the six
    // transforms on that block and meets the other shards in the cross stages (ntt.hip)
    bool part = false;
    uint32_t log_shards = 0;
    uint64_t nloc = 0;                                   // rows of a, b, c, h held here (= domainSize unless part)
    DevBuf<Fr> xb;                                       // exchange buffer of the cross stages: [3][shard][nloc / shards]
    Fr *abc_use = nullptr, *xb_use = nullptr;            // a|b|c blocks; exchange buffer (own, or the caller's: zk_shard_set_exchange)
    Fr *pk_use = nullptr;                                // caller's send/receive staging [GPU][poly][chunk] (all_to_all path only)
    Fr *peer_abc[8] = {nullptr}, *peer_xb[8] = {nullptr};   // inside one process: every shard's buffers (zk_multi_prover)
    bool have_peers = false;
    int phase_open = -1, phase_next = 0;                 // slot being submitted phase by phase (-1: none), next phase
    hipEvent_t ev_ext_in = nullptr, ev_ext_out = nullptr;
    uint32_t log_shards_chain() const { return part ? log_shards : 0; }
    // follow-up streams: partial merges + bucket reductions of MSM m run on tail[m] (m: 0 = A, 1 = B1, 2 = C,
    // 3 = H, 4 = B2); `tail_streams` distinct streams are shared among them (none = on the MSM's own stream)
    hipStream_t tail_pool[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t tail[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int tail_streams = 0;
    bool use_graph = false;     // replay captured graphs (ZKHIP_GRAPH=1; small unsharded-chain provers without timings)
    bool capturing = false;     // the phases are being recorded into a slot's graph, not executed
    bool batch_abc = false;     // MSM A, B1, C in one set of launches (small circuits; ZKHIP_BATCH_ABC=0/1 overrides)
    hipStream_t stream_fin = nullptr;                   // joins a proof's streams and copies its window sums to the host
    hipStream_t stream_h2d = nullptr;                   // witness uploads of host-witness proofs

    double timings[ZK_T_COUNT] = {0};
    uint32_t accum_launches = 0;
    uint64_t launches_at_front = 0, launches_last_proof = 0;     // hipcheck.hpp's launch counter around the last submitted proof

    ~zk_prover() {
        // proofs may still be in flight (submitted, never collected): drain before anything is released
        for (auto &x : extra)
            if (x && x->stream) { (void)hipStreamSynchronize(x->stream); (void)hipStreamSynchronize(x->stream2); }
        for (hipStream_t st : {stream_h2d, stream, stream2, tail_pool[0], tail_pool[1], tail_pool[2], tail_pool[3], tail_pool[4], stream_fin})
            if (st) (void)hipStreamSynchronize(st);
        if (ev_ext_in) (void)hipEventDestroy(ev_ext_in);
        if (ev_ext_out) (void)hipEventDestroy(ev_ext_out);
        if (stream_h2d) (void)hipStreamDestroy(stream_h2d);
        if (stream_fin) (void)hipStreamDestroy(stream_fin);
        for (hipStream_t st : tail_pool)
            if (st) (void)hipStreamDestroy(st);
        if (stream2 && stream2 != stream) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace zkp {

inline void need_device_count() {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw std::runtime_error("no HIP device available (libzkhip has no CPU fallback)");
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        HIP_TRY(hipSetDevice(dev));
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <class Fn>
int guarded(Fn fn) {
    try {
        fn();
        return 0;
    } catch (const HipError &e) {
        set_error(std::string("HIP failure: ") + e.what());
        return 2;
    } catch (const std::exception &e) {
        set_error(e.what());
        return 1;
    }
}

inline uint32_t ilog2_exact(uint64_t n) {
    uint32_t l = 0;
    while ((1ull << l) < n) l++;
    if ((1ull << l) != n) throw std::invalid_argument("domainSize is not a power of two");
    return l;
}

// ---- prover_create.hip
void prover_create(zk_prover **out, const zk_zkey_view *z, const zk_opts *o);
void alloc_slot(zk_prover *p, int i);                                       // device + pinned workspace of one in-flight proof
void ensure_witness_buffer(zk_prover *p, zk_prover::ProofSlot &q);         // the slot's HBM witness buffer and upload events

// ---- prover_pipeline.hip: the phases of one proof (see the comment above phase_front), submit / collect
struct BatchIn {           // several witnesses for one submission of a batch prover (r32s / s32s: count x 32 bytes, or NULL = drawn at collect)
    const uint8_t *const *wtns;
    uint32_t count;
    const uint8_t *r32s, *s32s;
};
void stage_job_run(void *arg);              // host function: pageable witness -> pinned staging (StageJob)
int phase_front(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, const uint8_t *r32, const uint8_t *s32, hipEvent_t src_ready = nullptr,
                const BatchIn *bi = nullptr);
void phase_cross_push(zk_prover *p, hipEvent_t pushed);
void phase_cross_run(zk_prover *p, bool inverse, hipEvent_t done);
void phase_local(zk_prover *p);
void phase_back(zk_prover *p);
struct PhaseAbort {        // a failed phase must not leave the prover wedged in "being submitted"
    zk_prover *p;
    bool armed = true;
    ~PhaseAbort() { if (armed) p->phase_open = -1; }
};
struct SubmittedRS {
    uint8_t r32[32], s32[32];
    bool have_r = false, have_s = false;
};
// `direct` (unsharded provers): assemble the proof straight from the window sums instead of filling `out`;
// (r32, s32) given by the caller override the ones captured at submit (synchronous zk_prove).
struct DirectProof {
    zk_proof *out;              // `count` proofs (1 unless the submission was a batch)
    const uint8_t *r32, *s32;
    bool use_submitted;
    uint32_t count = 1;
};
void collect_sums(zk_prover *p, zk_msm_sums *out, SubmittedRS *rs = nullptr, const DirectProof *direct = nullptr);
void prove_finish(zk_prover *p, const zk_msm_sums *parts, uint32_t nparts, const uint8_t *r32, const uint8_t *s32, zk_proof *out);

}   // namespace zkp
