// C-ABI of libzkhip (include/zkhip.h): the MI355X replacement of
// Groth16::makeProver / Prover::prove (reference src/groth16.cpp:9-254).
//
// zk_prover_create does the one-off work (CSR of the coefficient records, point tables and
// twiddles resident in HBM); zk_prove* run the per-proof pipeline:
//   SpMV -> c=a.b -> 3x(DIF iNTT, coset*1/n, DIT NTT) -> h -> digits/sort(w), digits/sort(h)
//   -> bucket accumulation A,B1,C,H (G1) and B2 (G2) -> bucket reduce -> host Horner + assembly.
// No CPU fallback exists: every entry point fails if HIP does.
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

namespace {

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
static Slice shard_slice(uint64_t n, uint32_t idx, uint32_t cnt) {
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
    void alloc(uint64_t n_, uint32_t window_bits, bool precomp = false, uint32_t batch = 1) {
        plan = make_msm_plan(n_ ? n_ : 1, window_bits, precomp, batch);
        n = n_ * (batch > 1 ? batch : 1);
        // sort entries are 32-bit (bit 31 = digit sign): positions n*W and, with window-precomputed
        // tables, table rows j*n + i must stay below 2^32 / 2^31
        if ((n ? n : 1) * plan.W >= (1ull << 32)) throw std::invalid_argument("MSM too large: n * windows >= 2^32 sort entries");
        if ((n ? n : 1) * (precomp ? plan.W : 1) >= (1ull << 31)) throw std::invalid_argument("MSM too large: table rows >= 2^31");
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
    void run(const Fr *scalars, hipStream_t s) {
        MsmSortBufs b{offsets.p, entries.p, counts.p, starts.p, codes.p, val.p, bin_counts.p, bin_starts.p, lo.p};
        launch_msm_sort(b, scalars, n, plan, s);
    }
};

}   // namespace

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
    bool precomp = false;      // window-precomputed tables (ZK_FLAG_PRECOMP): tables hold W rows of n points
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
    uint32_t batch = 1;         // opts.batch: witnesses proved by one submission (one set of kernel launches)
    // Small circuits: a proof is ~60 launches of kernels that each fill a tenth of the chip and wait on a
    // serial chain of point additions, so throughput comes from running SEVERAL PROOFS' kernels at once.
    // Consecutive proofs on the same streams cannot (stream order); `lanes` independent sets of
    // {stream, stream2, a|b|c, h, sort(h) buffers} can: slot i uses lane i % lanes.  Lane 0 is the prover's
    // own streams and buffers above; the extra lanes run their follow-up kernels and the final copies on
    // their own two streams (hardware queues are few: csrc/prover.hip, GPU_MAX_HW_QUEUES).
    struct LaneExtra {
        hipStream_t stream = nullptr, stream2 = nullptr;
        DevBuf<Fr> abc, h;
        SortBufs sort_h;
        // the streams (a hardware queue each: ~8 ms to create) and the buffers (~1 GiB per lane at 2^22) appear the first
        // time a proof runs on the lane, like the proof slots: the one-shot CLI and the shards of a sharded proof never
        // use more than lane 0
        uint64_t n_abc = 0, n_h = 0, nh_sort = 0;
        uint32_t wbits = 0, batch = 1;
        bool precomp = false, one_stream = false;
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

namespace {

void need_device_count() {
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

uint32_t ilog2_exact(uint64_t n) {
    uint32_t l = 0;
    while ((1ull << l) < n) l++;
    if ((1ull << l) != n) throw std::invalid_argument("domainSize is not a power of two");
    return l;
}

// Device and pinned-host workspace of one in-flight proof.
static void alloc_slot(zk_prover *p, int i) {
    zk_prover::ProofSlot &q = p->slot[i];
    if (q.allocated) return;
    const uint64_t nv = p->sv.size();
    q.sort_w.alloc(nv, p->wbits, p->precomp, p->batch);
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
    if (p->batch > 1 && (!(p->flags & ZK_FLAG_PRECOMP) || p->shard_count != 1 || (p->flags & ZK_FLAG_PARTITIONED_CHAIN)))
        throw std::invalid_argument("opts.batch needs ZK_FLAG_PRECOMP on an unsharded prover");
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
        const bool mid_size_unsharded = p->shard_count == 1 && p->sv.size() >= (1u << 20) && p->sv.size() < (1u << 22);
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
    p->sort_h.alloc(nh, wbits, p->precomp, p->batch);
    alloc_slot(p.get(), 0);
    // with window pre-computation a table holds W rows: row j = 2^(c*j) * P (msm.hip)
    const uint64_t rows_w = p->precomp ? p->slot[0].sort_w.plan.W : 1, rows_h = p->precomp ? p->sort_h.plan.W : 1;
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
        const uint64_t tw = (uint64_t)(plan_w.W - 1) * (nv ? nv : 1), th = (uint64_t)(p->sort_h.plan.W - 1) * (nh ? nh : 1);
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
            x->precomp = p->precomp;
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
static void stage_job_run(void *arg) {
    const StageJob *j = (const StageJob *)arg;
    stage_pool().copy(j->dst, j->src, j->bytes);
}
// the slot's HBM witness buffer and the events of its upload (host-witness proofs)
static void ensure_witness_buffer(zk_prover *p, zk_prover::ProofSlot &q) {
    if (!q.wtns_dev.p) q.wtns_dev.alloc((uint64_t)p->nVars * p->batch);
    if (!q.ev_h2d) HIP_TRY(hipEventCreateWithFlags(&q.ev_h2d, hipEventDisableTiming));
    if (!q.ev_h2d_start) HIP_TRY(hipEventCreate(&q.ev_h2d_start));
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
    AccumTail tail_of(int m) const { AccumTail t; t.stream = tail[m]; t.l1_done = q.ev_l1[m]; t.buckets_zeroed = q.zeroed; return t; }
    hipStream_t after(int m, hipStream_t own) const { return tail[m] ? tail[m] : own; }
    NttTables tables() const { return NttTables{p->logn, p->tw_fwd.p, p->tw_inv.p, p->tw_coset.p, p->tw_ninv.p}; }
};

// several witnesses for one submission of a batch prover (r32s / s32s: count x 32 bytes, or NULL = drawn at collect)
struct BatchIn {
    const uint8_t *const *wtns;
    uint32_t count;
    const uint8_t *r32s, *s32s;
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
    // 2^17 2.98 -> 2.58, 2^18 4.46 -> 4.18, 2^19 7.52 -> 7.35; at 2^22 it LOSES 0.2-2 ms (r04am: the merges then run beside the
    // chip-filling A|B1|C launch, which they slow down more than their own 0.5 ms), hence the size limit.
    static const uint32_t g2_aside_maxlog = [] { const char *e = probe_env("ZKHIP_G2_ASIDE_MAXLOG"); return e ? (uint32_t)atoi(e) : 19u; }();
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

int phase_front(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, const uint8_t *r32, const uint8_t *s32, hipEvent_t src_ready = nullptr,
                const BatchIn *bi = nullptr) {
    DeviceGuard g(p->device);
    if (p->in_flight >= ZK_MAX_IN_FLIGHT) throw std::invalid_argument("too many proofs in flight (ZK_MAX_IN_FLIGHT): collect one first");
    if (p->phase_open >= 0) throw std::invalid_argument("a proof is still being submitted phase by phase");
    // nothing in flight: start over at slot 0 — a caller that proves one at a time (zk_prove, the reference's contract) then
    // lives in ONE slot on lane 0 instead of walking through all eight (each with ~1 GiB of buffers at 2^22, a pinned
    // staging copy and, per lane, two streams its first use has to create: the first eight proofs of a prover paid
    // 10-20 ms each for that).  (Not while a graph is being captured: submit_graph has chosen the slot.)
    if (p->in_flight == 0 && !p->capturing) p->next_submit = p->next_collect = 0;
    const int si = (int)(p->next_submit % ZK_MAX_IN_FLIGHT);
    alloc_slot(p, si);
    // Follow-up streams only for a proof that is submitted while another one is in flight: they let the next level-1 launch
    // start beside the merges and reductions of the previous MSM — worth 2 % of a sharded prover's period with two in flight —
    // but every hop to them is an event across hardware queues, and a LONE proof is faster without (same box, with / without:
    // 2^22 synchronous 39.0-39.8 / 37.9-38.0 ms, 2^20 13.3 / 12.4 ms; rank-0 share of 8 shards one at a time 7.8-8.0 / 6.5 ms).
    // (A captured graph keeps whatever its slot recorded.)
    if (!p->capturing) p->slot[si].use_tails = p->in_flight > 0 || p->use_graph;
    PhaseCtx c(p, si);
    zk_prover::ProofSlot &q = c.q;
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
        launch_ntt_coset_pair(c.abc, nl, 3 * c.q.count, p->pair, c.s);      // inverse, coset shift * 1/n, forward: nttpair.hip
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
    q.busy = true;
    q.via_graph = false;
    p->phase_open = -1;
    p->next_submit++;
    p->in_flight++;
}

// a failed phase must not leave the prover wedged in "being submitted"
struct PhaseAbort {
    zk_prover *p;
    bool armed = true;
    ~PhaseAbort() { if (armed) p->phase_open = -1; }
};

}   // namespace

// Graph path (small circuits): a proof's device work — everything behind the witness upload — is recorded once
// per slot by running the same three phases under stream capture, then replayed with one hipGraphLaunch on the
// lane's stream 1.  The upload stays outside (its source changes with every proof), and so does the event a
// collect waits for.
static void submit_graph(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, const uint8_t *r32, const uint8_t *s32) {
    if (p->in_flight >= ZK_MAX_IN_FLIGHT) throw std::invalid_argument("too many proofs in flight (ZK_MAX_IN_FLIGHT): collect one first");
    if (p->phase_open >= 0) throw std::invalid_argument("a proof is still being submitted phase by phase");
    if (p->in_flight == 0) p->next_submit = p->next_collect = 0;      // (see phase_front)
    const int si = (int)(p->next_submit % ZK_MAX_IN_FLIGHT);
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
static void collect_sums(zk_prover *p, zk_msm_sums *out, SubmittedRS *rs = nullptr, const DirectProof *direct = nullptr) {
    std::lock_guard<std::mutex> ck(p->cmtx);
    zk_prover::ProofSlot *qp;
    {
        std::lock_guard<std::mutex> lk(p->mtx);
        if (!p->in_flight) throw std::invalid_argument("no proof in flight");
        qp = &p->slot[p->next_collect % ZK_MAX_IN_FLIGHT];
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
    if (Ww == 1 && Wh == 1) {          // window-precomputed tables: one sum per MSM, nothing to run in parallel
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
void prove_msm(zk_prover *p, const Fr *d_wtns, const uint8_t *h_wtns, zk_msm_sums *out, const DirectProof *direct = nullptr) {
    std::lock_guard<std::mutex> one(p->sync_mtx);
    {
        std::lock_guard<std::mutex> lk(p->mtx);
        if (p->in_flight) throw std::invalid_argument("asynchronous proofs in flight: collect them first");
        submit_locked(p, d_wtns, h_wtns, nullptr, nullptr);
    }
    collect_sums(p, out, nullptr, direct);
}

void prove_finish(zk_prover *p, const zk_msm_sums *parts, uint32_t nparts, const uint8_t *r32, const uint8_t *s32, zk_proof *out) {
    if (zk_assemble(p->vk_alpha1, p->vk_beta1, p->vk_beta2, p->vk_delta1, p->vk_delta2, parts, nparts, r32, s32, out))
        throw std::runtime_error(get_error());
}

}   // namespace

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

int zk_prover_reserve(zk_prover *p, uint32_t in_flight, uint32_t host_witnesses) {
    return guarded([&] {
        if (!p) throw std::invalid_argument("null argument");
        if (in_flight > ZK_MAX_IN_FLIGHT) throw std::invalid_argument("in_flight > ZK_MAX_IN_FLIGHT");
        std::lock_guard<std::mutex> lk(p->mtx);
        DeviceGuard g(p->device);
        // a pipeline walks the whole ring of slots (slot i runs on lane i % lanes); one proof at a time lives in slot 0
        const int nslots = in_flight >= 2 ? ZK_MAX_IN_FLIGHT : 1;
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

int zk_prover_timings(zk_prover *p, double *ms, uint32_t n) {
    return guarded([&] {
        if (!p || !ms) throw std::invalid_argument("null argument");
        if (!(p->flags & ZK_FLAG_TIMINGS)) throw std::invalid_argument("prover created without ZK_FLAG_TIMINGS");
        for (uint32_t i = 0; i < n && i < ZK_T_COUNT; i++) ms[i] = p->timings[i];
    });
}

}   // extern "C"

// ------------------------------------------------------------------ one proof on several GPUs
// (a) zk_shard_*: one process per GPU (torch.distributed / RCCL): the caller owns the exchange of the
//     chain's blocks — four all_to_all per proof on buffers it registered — and drives the phases.
// (b) zk_multi_prover: all GPUs of the node in ONE process (what the reference's CLI and server are):
//     one shard prover per device, phases enqueued device by device, blocks exchanged by peer writes
//     over xGMI, cross-device ordering by events (hipStreamWaitEvent across devices).
namespace {

// stream 1 of the prover <-> the caller's stream (on which its collectives are ordered)
// (a NULL handle is the default stream, which is what torch.cuda.current_stream() is unless the caller
// switched streams: it must be ordered like any other)
void ext_in(zk_prover *p, void *stream) {
    HIP_TRY(hipEventRecord(p->ev_ext_in, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_ext_in, 0));
}
void ext_out(zk_prover *p, void *stream) {
    HIP_TRY(hipEventRecord(p->ev_ext_out, p->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->ev_ext_out, 0));
}

}   // namespace

struct zk_multi_prover {
    std::vector<zk_prover *> shard;
    std::vector<hipEvent_t> ev[4];       // per shard: chunks pushed (DIF), cross DIF done, chunks pushed (DIT), cross DIT done
    uint8_t *stage_pin[ZK_MAX_IN_FLIGHT] = {nullptr};
    hipEvent_t ev_staged[ZK_MAX_IN_FLIGHT] = {nullptr};
    StageJob stage[ZK_MAX_IN_FLIGHT];
    uint64_t submitted = 0;
    bool part = false;
    std::mutex mtx, cmtx, sync_mtx;      // submit / collect / one synchronous call at a time (as in zk_prover)
    ~zk_multi_prover() {
        for (zk_prover *q : shard) zk_prover_destroy(q);        // drains every stream first
        for (auto &v : ev) for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e);
        for (auto &b : stage_pin) if (b) (void)hipHostFree(b);
        for (auto &e : ev_staged) if (e) (void)hipEventDestroy(e);
    }
};

namespace {

void multi_create(zk_multi_prover **out, const zk_zkey_view *z, const int32_t *devices, uint32_t nd, const zk_opts *o) {
    if (!out || !z || !devices || nd == 0) throw std::invalid_argument("null argument");
    if (nd > 8) throw std::invalid_argument("at most 8 devices");
    need_device_count();
    std::unique_ptr<zk_multi_prover> mp(new zk_multi_prover());
    uint32_t lg = 0;
    while ((1u << lg) < nd) lg++;
    uint32_t logn = 0;
    while ((1ull << logn) < z->domainSize) logn++;
    bool can_part = nd > 1 && (1u << lg) == nd && logn >= 2 * lg && !getenv("ZKHIP_REPLICATED_CHAIN");
    // the partitioned chain writes into its peers' buffers: every pair of distinct devices must be able to map each other
    // (xGMI inside a node).  Where one cannot, the chain stays replicated (partial sums only travel through the host).
    bool peers_ok = true;
    for (uint32_t a = 0; a < nd && peers_ok; a++)
        for (uint32_t b = 0; b < nd && peers_ok; b++) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
            if (!can) peers_ok = false;
        }
    if (!peers_ok) can_part = false;
    mp->part = can_part;
    for (uint32_t g = 0; g < nd; g++) {
        zk_opts so;
        memset(&so, 0, sizeof so);
        so.device = devices[g];
        so.shard_index = g;
        so.shard_count = nd;
        so.window_bits = o ? o->window_bits : 0;
        so.flags = (o ? o->flags : 0) & ~ZK_FLAG_PARTITIONED_CHAIN;
        if (can_part) so.flags |= ZK_FLAG_PARTITIONED_CHAIN;
        zk_prover *q = nullptr;
        prover_create(&q, z, &so);
        mp->shard.push_back(q);
    }
    // peer access between every pair of distinct devices (xGMI inside a node); only the partitioned chain needs it
    for (uint32_t a = 0; a < nd && can_part; a++)
        for (uint32_t b = 0; b < nd; b++) {
            if (devices[a] == devices[b]) continue;
            DeviceGuard g(devices[a]);
            hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else HIP_TRY(e);
        }
    for (uint32_t a = 0; a < nd; a++) {
        zk_prover *q = mp->shard[a];
        for (uint32_t b = 0; b < nd; b++) {
            q->peer_abc[b] = mp->shard[b]->abc_use;
            q->peer_xb[b] = mp->shard[b]->xb_use;
        }
        q->have_peers = can_part;
        DeviceGuard g(q->device);
        for (auto &v : mp->ev) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            v.push_back(e);
        }
    }
    {
        DeviceGuard g(mp->shard[0]->device);
        for (auto &e : mp->ev_staged) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = mp.release();
}

// every stream-1 waits for the given event of EVERY shard (the all-to-all dependency of a cross step)
void wait_all(zk_multi_prover *mp, int which) {
    for (zk_prover *q : mp->shard) {
        DeviceGuard g(q->device);
        for (size_t b = 0; b < mp->shard.size(); b++) HIP_TRY(hipStreamWaitEvent(q->stream, mp->ev[which][b], 0));
    }
}

void multi_submit(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32) {
    const size_t G = mp->shard.size();
    std::vector<std::unique_lock<std::mutex>> locks;
    for (zk_prover *q : mp->shard) locks.emplace_back(q->mtx);
    zk_prover *p0 = mp->shard[0];
    if (p0->in_flight >= ZK_MAX_IN_FLIGHT) throw std::invalid_argument("too many proofs in flight (ZK_MAX_IN_FLIGHT): collect one first");
    // the witness goes to every GPU (each needs all of it for its rows of A.w / B.w): staged ONCE into
    // pinned memory (host function on shard 0's upload stream), then G DMA copies
    const uint8_t *src = wtns;
    hipEvent_t ready = nullptr;
    {
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, wtns) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (!pinned && G > 1) {
            const int k = (int)(mp->submitted % ZK_MAX_IN_FLIGHT);
            const size_t bytes = (size_t)p0->nVars * 32;
            DeviceGuard g(p0->device);
            if (!mp->stage_pin[k]) HIP_TRY(hipHostMalloc((void **)&mp->stage_pin[k], bytes, hipHostMallocPortable));
            mp->stage[k] = StageJob{mp->stage_pin[k], wtns, bytes};
            HIP_TRY(hipLaunchHostFunc(p0->stream_h2d, stage_job_run, &mp->stage[k]));
            HIP_TRY(hipEventRecord(mp->ev_staged[k], p0->stream_h2d));
            src = mp->stage_pin[k];
            ready = mp->ev_staged[k];
        }
    }
    std::vector<PhaseAbort> guards;
    guards.reserve(G);
    for (zk_prover *q : mp->shard) guards.push_back(PhaseAbort{q});
    if (!mp->part) {
        for (zk_prover *q : mp->shard) {
            phase_front(q, nullptr, src, r32, s32, ready);
            phase_local(q);
            phase_back(q);
        }
    } else {
        for (size_t a = 0; a < G; a++) {
            phase_front(mp->shard[a], nullptr, src, r32, s32, ready);
            phase_cross_push(mp->shard[a], mp->ev[0][a]);
        }
        wait_all(mp, 0);
        for (size_t a = 0; a < G; a++) phase_cross_run(mp->shard[a], true, mp->ev[1][a]);
        wait_all(mp, 1);
        for (size_t a = 0; a < G; a++) {
            phase_local(mp->shard[a]);
            phase_cross_push(mp->shard[a], mp->ev[2][a]);
        }
        wait_all(mp, 2);
        for (size_t a = 0; a < G; a++) phase_cross_run(mp->shard[a], false, mp->ev[3][a]);
        wait_all(mp, 3);
        for (size_t a = 0; a < G; a++) phase_back(mp->shard[a]);
    }
    for (auto &gd : guards) gd.armed = false;
    mp->submitted++;
}

void multi_collect(zk_multi_prover *mp, zk_proof *out) {
    const size_t G = mp->shard.size();
    std::vector<zk_msm_sums> sums(G);
    SubmittedRS rs;
    // EVERY shard's oldest submission is retired, whatever one of them throws: a shard left behind would pair its partial
    // sums with the next proof's on every later collect
    std::exception_ptr first;
    for (size_t a = 0; a < G; a++) {
        try {
            collect_sums(mp->shard[a], &sums[a], a == 0 ? &rs : nullptr);
        } catch (...) {
            if (!first) first = std::current_exception();
        }
    }
    if (first) std::rethrow_exception(first);
    prove_finish(mp->shard[0], sums.data(), (uint32_t)G, rs.have_r ? rs.r32 : nullptr, rs.have_s ? rs.s32 : nullptr, out);
}

}   // namespace

extern "C" {

int zk_multi_prover_create(zk_multi_prover **out, const zk_zkey_view *zkey, const int32_t *devices, uint32_t n_devices, const zk_opts *opts) {
    return guarded([&] { multi_create(out, zkey, devices, n_devices, opts); });
}

void zk_multi_prover_destroy(zk_multi_prover *mp) { delete mp; }

int zk_multi_prove_submit(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32) {
    return guarded([&] {
        if (!mp || !wtns) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(mp->mtx);
        multi_submit(mp, wtns, r32, s32);
    });
}

int zk_multi_prove_collect(zk_multi_prover *mp, zk_proof *out) {
    return guarded([&] {
        if (!mp || !out) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(mp->cmtx);          // submissions (mp->mtx) go on meanwhile
        multi_collect(mp, out);
    });
}

int zk_multi_prove(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out) {
    return guarded([&] {
        if (!mp || !wtns || !out) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> one(mp->sync_mtx);
        {
            std::lock_guard<std::mutex> lk(mp->mtx);
            if (mp->shard[0]->in_flight) throw std::invalid_argument("asynchronous proofs in flight: collect them first");
            multi_submit(mp, wtns, r32, s32);
        }
        std::lock_guard<std::mutex> lk(mp->cmtx);
        multi_collect(mp, out);
    });
}

int zk_multi_prover_info(zk_multi_prover *mp, uint32_t *n_shards, uint32_t *chain_partitioned) {
    return guarded([&] {
        if (!mp) throw std::invalid_argument("null argument");
        if (n_shards) *n_shards = (uint32_t)mp->shard.size();
        if (chain_partitioned) *chain_partitioned = mp->part ? 1u : 0u;
    });
}

int zk_shard_info(zk_prover *p, uint64_t *block_elems, uint32_t *chain_partitioned) {
    return guarded([&] {
        if (!p) throw std::invalid_argument("null argument");
        if (block_elems) *block_elems = p->nloc;
        if (chain_partitioned) *chain_partitioned = p->part ? 1u : 0u;
    });
}

int zk_shard_set_exchange(zk_prover *p, void *d_send, void *d_recv) {
    return guarded([&] {
        if (!p || !d_send || !d_recv) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(p->mtx);
        if (!p->part) throw std::invalid_argument("prover was not created with ZK_FLAG_PARTITIONED_CHAIN");
        if (p->in_flight || p->phase_open >= 0) throw std::invalid_argument("proofs in flight");
        p->pk_use = (Fr *)d_send;
        p->xb_use = (Fr *)d_recv;
        p->xb.release();
    });
}

int zk_shard_begin(zk_prover *p, const uint8_t *wtns, const void *d_wtns, const uint8_t *r32, const uint8_t *s32, void *stream) {
    return guarded([&] {
        if (!p || (!wtns == !d_wtns)) throw std::invalid_argument("exactly one of wtns / d_wtns");
        std::lock_guard<std::mutex> lk(p->mtx);
        if (!p->part) throw std::invalid_argument("prover was not created with ZK_FLAG_PARTITIONED_CHAIN");
        DeviceGuard g(p->device);
        PhaseAbort guard{p};
        ext_in(p, stream);
        phase_front(p, (const Fr *)d_wtns, wtns, r32, s32);
        ext_out(p, stream);
        guard.armed = false;
    });
}

int zk_shard_step(zk_prover *p, int step, void *stream) {
    return guarded([&] {
        if (!p) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(p->mtx);
        DeviceGuard g(p->device);
        PhaseAbort guard{p};
        ext_in(p, stream);
        switch (step) {
        case ZK_STEP_CROSS_INVERSE: phase_cross_run(p, true, nullptr); break;
        case ZK_STEP_LOCAL: phase_local(p); break;
        case ZK_STEP_CROSS_FORWARD: phase_cross_run(p, false, nullptr); break;
        case ZK_STEP_FINISH: phase_back(p); break;
        default: throw std::invalid_argument("unknown step");
        }
        ext_out(p, stream);
        guard.armed = false;
    });
}

}   // extern "C"

// ------------------------------------------------------------------ operator level
static void need_device() { need_device_count(); }

template <class F>
static void mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n, void (*launch)(F *, const F *, const F *, uint64_t, hipStream_t)) {
    need_device();
    if (!n) return;
    DevBuf<F> da, db;
    da.alloc(n);
    db.alloc(n);
    da.upload(a, n, 0);
    db.upload(b, n, 0);
    launch(da.p, da.p, db.p, n, 0);
    HIP_TRY(hipMemcpy(out, da.p, n * 32, hipMemcpyDeviceToHost));
}

struct Tables {
    DevBuf<TwEntry> fwd, inv;
    DevBuf<Fr> coset, ninv;
    NttTables t;
    void build(uint32_t logn) {
        uint64_t n = 1ull << logn;
        fwd.alloc(n > 1 ? n / 2 : 1);
        inv.alloc(n > 1 ? n / 2 : 1);
        coset.alloc(n);
        ninv.alloc(1);
        launch_ntt_build_tables(fwd.p, inv.p, coset.p, ninv.p, logn, 0);
        t = NttTables{logn, fwd.p, inv.p, coset.p, ninv.p};
    }
};

template <class AffT, class XT, class AccT>
static void msm_generic(uint8_t *out, const uint8_t *bases, const uint8_t *scalars, uint64_t n, bool g2);

extern "C" {

int zk_fr_mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n) {
    return guarded([&] { mul_vec<Fr>(out, a, b, n, launch_fr_mul_vec); });
}
int zk_fq_mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n) {
    return guarded([&] { mul_vec<Fq>(out, a, b, n, launch_fq_mul_vec); });
}

// a = A.w, b = B.w over the packed coefficient records — the accumulation loop of src/groth16.cpp:62-85
// as an operator: same records (section 4 incl. its u32 count), same witness form, a and b come back
// in the reference's Montgomery form (what its a[] / b[] arrays hold after line 85).
int zk_fr_coef_accumulate(uint8_t *a, uint8_t *b, const void *coefs, uint64_t nCoefs, uint32_t domainSize, const uint8_t *wtns, uint32_t nVars) {
    return guarded([&] {
        need_device();
        if (!a || !b || !coefs || !wtns || !domainSize || !nVars) throw std::invalid_argument("null argument");
        if (nCoefs >= (1ull << 32)) throw std::invalid_argument("nCoefs >= 2^32 is not supported");
        const uint32_t rows = 2 * domainSize;
        DevBuf<uint8_t> raw;
        DevBuf<uint32_t> cursor, err, rowptr, col;
        DevBuf<Fr> val, w, ab;
        raw.alloc(nCoefs ? nCoefs * 44 : 4);
        cursor.alloc(rows);
        err.alloc(1);
        rowptr.alloc((size_t)rows + 1 + msm_scan_extra_words(rows));
        col.alloc(nCoefs ? nCoefs : 1);
        val.alloc(nCoefs ? nCoefs : 1);
        w.alloc(nVars);
        ab.alloc(3 * (size_t)domainSize);
        if (nCoefs) HIP_TRY(hipMemcpy(raw.p, (const uint8_t *)coefs + 4, nCoefs * 44, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(w.p, wtns, (size_t)nVars * 32, hipMemcpyHostToDevice));
        launch_csr_build(rowptr.p, col.p, val.p, cursor.p, err.p, raw.p, nCoefs, domainSize, nVars, 0, domainSize, 0);
        launch_fr_to_internal(val.p, nCoefs, 2, 0);
        uint32_t bad = 0;
        HIP_TRY(hipMemcpy(&bad, err.p, 4, hipMemcpyDeviceToHost));
        if (bad) throw std::invalid_argument("zkey coefficient record out of range");
        CsrDev csr{rowptr.p, col.p, val.p};
        launch_spmv_abc(ab.p, ab.p + domainSize, ab.p + 2 * (size_t)domainSize, csr, w.p, domainSize, 0);
        launch_fr_from_internal(ab.p, 2 * (size_t)domainSize, 0);
        HIP_TRY(hipMemcpy(a, ab.p, (size_t)domainSize * 32, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b, ab.p + domainSize, (size_t)domainSize * 32, hipMemcpyDeviceToHost));
    });
}

// Tables of the stand-alone transforms, kept per device for the size used last (2 n field elements: a caller that
// transforms many vectors of one size — zkgen, the KATs — builds them once).
static std::mutex g_plain_mtx;
static std::map<int, std::unique_ptr<NttPair>> &g_plain_tables = *new std::map<int, std::unique_ptr<NttPair>>();      // (never destroyed: no hipFree behind the runtime's own exit handlers)

int zk_fr_ntt(uint8_t *data, uint64_t n, int inverse) {
    return guarded([&] {
        need_device();
        uint32_t logn = ilog2_exact(n);
        if (logn > 28) throw std::invalid_argument("n exceeds 2^28");
        DevBuf<Fr> d;
        d.alloc(n);
        d.upload(data, n, 0);
        launch_fr_to_internal(d.p, n, 1, 0);          // x*2^256 -> x*2^261
        if (ntt_pair_supported(logn) && !probe_env("ZKHIP_NTT_RADIX2")) {
            // the proof path's own passes (nttpair.hip: register radix-8 butterflies, clean sub-transforms), the bit reversal
            // folded into the middle pass's addressing: no permutation pass
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            std::lock_guard<std::mutex> lk(g_plain_mtx);
            std::unique_ptr<NttPair> &tp = g_plain_tables[dev];
            if (!tp || tp->L != logn) {
                tp.reset(new NttPair());
                tp->build(logn, logn, 0, 0, /*plain=*/true);
            }
            DevBuf<Fr> d2;
            d2.alloc(n);
            launch_ntt_plain(d2.p, d.p, n, 1, *tp, inverse != 0, 0);
            launch_fr_from_internal(d2.p, n, 0);
            HIP_TRY(hipMemcpy(data, d2.p, n * 32, hipMemcpyDeviceToHost));
            return;
        }
        // sizes the pipeline does not take (n < 8, n = 2^28): radix-2 passes + a permutation pass (ntt.hip)
        Tables tb;
        tb.build(logn);
        if (inverse) {
            launch_ntt_dif_inverse(d.p, n, 1, tb.t, 0);
            launch_bitrev_permute(d.p, logn, 0);
            launch_fr_scale_const(d.p, tb.ninv.p, n, 0);
        } else {
            launch_bitrev_permute(d.p, logn, 0);
            launch_ntt_dit_forward(d.p, n, 1, tb.t, 0);
        }
        launch_fr_from_internal(d.p, n, 0);
        HIP_TRY(hipMemcpy(data, d.p, n * 32, hipMemcpyDeviceToHost));
    });
}

int zk_fr_abc_to_h(uint8_t *h, const uint8_t *a, const uint8_t *b, uint64_t n) {
    return guarded([&] {
        need_device();
        uint32_t logn = ilog2_exact(n);
        if (logn > 27) throw std::invalid_argument("n exceeds 2^27");
        Tables tb;
        tb.build(logn);
        DevBuf<Fr> abc, hh;
        abc.alloc(3 * n);
        hh.alloc(n);
        HIP_TRY(hipMemcpy(abc.p, a, n * 32, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(abc.p + n, b, n * 32, hipMemcpyHostToDevice));
        launch_fr_mul_vec(abc.p + 2 * n, abc.p, abc.p + n, n, 0);      // c = a o b in the reference's form
        launch_fr_to_internal(abc.p, 3 * n, 1, 0);
        if (ntt_pair_supported(logn) && !probe_env("ZKHIP_NTT_RADIX2")) {      // the proof path's pipeline (nttpair.hip)
            NttPair pr;
            pr.build(logn, logn, 0, 0);
            launch_ntt_coset_pair(abc.p, n, 3, pr, 0);
            HIP_TRY(hipStreamSynchronize(0));
        } else {
            launch_ntt_dif_inverse(abc.p, n, 3, tb.t, 0);
            launch_ntt_dit_forward(abc.p, n, 3, tb.t, 0, tb.coset.p);
        }
        launch_abc_to_h(hh.p, abc.p, abc.p + n, abc.p + 2 * n, n, 0);
        HIP_TRY(hipMemcpy(h, hh.p, n * 32, hipMemcpyDeviceToHost));
    });
}

}   // extern "C"

template <class AffT, class XT, class AccT>
static void msm_generic(uint8_t *out, const uint8_t *bases, const uint8_t *scalars, uint64_t n, bool g2) {
    need_device();
    if (n >= (1ull << 31)) throw std::invalid_argument("n too large");
    if (n == 0) {
        memset(out, 0, sizeof(AffT));
        return;
    }
    DevBuf<AffT> pts;
    DevBuf<Fr> sc;
    pts.alloc(n);
    sc.alloc(n);
    pts.upload(bases, n, 0);
    launch_fq_to_internal((Fq *)pts.p, n * (sizeof(AffT) / 32), 0);
    sc.upload(scalars, n, 0);
    SortBufs sb;
    sb.alloc(n, 0);
    sb.run(sc.p, 0);
    DevBuf<AccT> buckets, scratch, ws;
    DevBuf<XT> wsum;
    DevBuf<uint32_t> wkey, wflag;
    const uint64_t emax = sb.max_entries(), slots = msm_accum_workspace_slots(emax);
    ws.alloc(slots);
    wkey.alloc(slots);
    wflag.alloc(slots);
    buckets.alloc(sb.total_buckets());
    scratch.alloc(msm_reduce_scratch_points(1, sb.plan));
    const uint32_t rc = msm_wsum_rc(sb.plan);
    wsum.alloc((uint64_t)sb.plan.sets * rc);
    std::vector<uint8_t> w((size_t)sb.plan.sets * rc * sizeof(XT));
    if constexpr (sizeof(AffT) == 64) {
        launch_msm_accum_g1((G1Acc *)buckets.p, sb.offsets.p, sb.entries.p, (const G1Affine *)pts.p, 0, 0, sb.total_buckets(), emax, (G1Acc *)ws.p, wkey.p, wflag.p, 0);
        launch_msm_reduce_g1((G1XYZZ *)wsum.p, (G1Acc *)scratch.p, (const G1Acc *)buckets.p, 1, sb.plan, 0);
    } else {
        launch_msm_accum_g2((G2Acc *)buckets.p, sb.offsets.p, sb.entries.p, (const G2Affine *)pts.p, 0, 0, sb.total_buckets(), emax, (G2Acc *)ws.p, wkey.p, wflag.p, 0);
        launch_msm_reduce_g2((G2XYZZ *)wsum.p, (G2Acc *)scratch.p, (const G2Acc *)buckets.p, 1, sb.plan, 0);
    }
    HIP_TRY(hipMemcpy(w.data(), wsum.p, w.size(), hipMemcpyDeviceToHost));
    if (g2) HostTail::combine_windows_g2(w.data(), sb.plan.sets, sb.plan.c, rc, out);
    else HostTail::combine_windows_g1(w.data(), sb.plan.sets, sb.plan.c, rc, out);
}

template <class AffT, class XT, class FT>
static void synth_chain(uint8_t *out, uint64_t n, const uint8_t *p0, const uint8_t *q,
                        void (*launch)(AffT *, XT *, FT *, const AffT &, const AffT &, uint64_t, hipStream_t)) {
    need_device();
    if (!n) return;
    DevBuf<AffT> d_out;
    DevBuf<XT> d_tmp;
    DevBuf<FT> d_pref;
    d_out.alloc(n);
    d_tmp.alloc(n);
    d_pref.alloc(n);
    AffT P0, Q;
    memcpy(&P0, p0, sizeof(AffT));
    memcpy(&Q, q, sizeof(AffT));
    launch(d_out.p, d_tmp.p, d_pref.p, P0, Q, n, 0);
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(AffT), hipMemcpyDeviceToHost));
}

template <class AffT, class XT, class FT>
static void fixed_base_batch(uint8_t *out, const uint8_t *base, const uint8_t *scalars, uint64_t n,
                             void (*launch)(AffT *, XT *, FT *, const AffT &, const uint32_t *, uint64_t, hipStream_t)) {
    need_device();
    if (!n) return;
    if (!out || !base || !scalars) throw std::invalid_argument("null argument");
    DevBuf<AffT> d_out;
    DevBuf<XT> d_tmp;
    DevBuf<FT> d_pref;
    DevBuf<uint32_t> d_sc;
    d_out.alloc(n);
    d_tmp.alloc(n);
    d_pref.alloc(n);
    d_sc.alloc(n * 8);
    AffT B;
    memcpy(&B, base, sizeof(AffT));
    HIP_TRY(hipMemcpy(d_sc.p, scalars, n * 32, hipMemcpyHostToDevice));
    launch(d_out.p, d_tmp.p, d_pref.p, B, d_sc.p, n, 0);
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(AffT), hipMemcpyDeviceToHost));
}

extern "C" {

int zk_fixed_base_g1(uint8_t *out, const uint8_t base[64], const uint8_t *scalars, uint64_t n) {
    return guarded([&] { fixed_base_batch<G1Affine, G1XYZZ, Fq>(out, base, scalars, n, launch_fixed_base_g1); });
}
int zk_fixed_base_g2(uint8_t *out, const uint8_t base[128], const uint8_t *scalars, uint64_t n) {
    return guarded([&] { fixed_base_batch<G2Affine, G2XYZZ, Fq2>(out, base, scalars, n, launch_fixed_base_g2); });
}

int zk_synth_chain_g1(uint8_t *out, uint64_t n, const uint8_t p0[64], const uint8_t q[64]) {
    return guarded([&] { synth_chain<G1Affine, G1XYZZ, Fq>(out, n, p0, q, launch_chain_g1); });
}
int zk_synth_chain_g2(uint8_t *out, uint64_t n, const uint8_t p0[128], const uint8_t q[128]) {
    return guarded([&] { synth_chain<G2Affine, G2XYZZ, Fq2>(out, n, p0, q, launch_chain_g2); });
}

int zk_msm_g1(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, uint64_t n) {
    return guarded([&] { msm_generic<G1Affine, G1XYZZ, G1Acc>(out, bases, scalars, n, false); });
}
int zk_msm_g2(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, uint64_t n) {
    return guarded([&] { msm_generic<G2Affine, G2XYZZ, G2Acc>(out, bases, scalars, n, true); });
}

}   // extern "C"
