// BN254-Fr transforms that are NOT on the proof path of an unpartitioned prover (that one is nttpair.hip):
//   * k_ntt_cross / k_chunk_move : the log2(G) stages over the TOP index bits of a transform partitioned across G = 2/4/8
//     GPUs (one radix-G butterfly per block offset, full-size twiddle tables) and the block <-> exchange-buffer moves;
//   * k_ntt_pass (radix-2, one LDS round trip per stage, no bit-reversal pass: inverse = DIF, forward = DIT, the coset
//     table multiplied in as the first forward pass loads) : rounds 1-2's transform, kept behind the stand-alone operator
//     zk_fr_ntt (ffiasm FFT<Fr>::fft/ifft, call sites src/groth16.cpp:102,115,120,133,139,152) and for blocks of fewer
//     than eight elements;
//   * the pointwise helpers (abc -> h of src/groth16.cpp:158-163, Montgomery-radix conversion) and the radix-2 tables.
// Arithmetic: 9x29-bit signed limbs (field29.hpp); HBM keeps canonical 256-bit words in the 2^261 Montgomery form.
#include <stdlib.h>
#include "kernels.hpp"
#include "hipcheck.hpp"
#include "common.hpp"
#include "field29.hpp"

namespace zk {

// Workgroup shape of a pass.  The tile (2^11 elements = 72 KiB of LDS) sets how many stages one launch
// covers; the thread count sets what the pass can run BESIDE: a 512-thread workgroup needs two waves
// of 104 VGPRs on every SIMD of its CU at once, which never fits next to the resident bucket-accumulation
// waves (G1: 3 x 136 of the 512 registers, G2 split-lane: 2 x 200), so the chain only advanced in the
// gaps between MSM kernels; 256 threads = one wave per SIMD does fit (ZKHIP_NTT_THREADS / ZKHIP_NTT_TILE
// override, tuning aids).
#define NTT_MAX_TILE_LOG 11
static uint32_t ntt_threads() {
    static const uint32_t v = [] { const char *e = probe_env("ZKHIP_NTT_THREADS"); uint32_t t = e ? (uint32_t)atoi(e) : 256u; return t == 512u ? 512u : 256u; }();
    return v;
}
static uint32_t ntt_tile_log() {
    static const uint32_t v = [] { const char *e = probe_env("ZKHIP_NTT_TILE"); uint32_t t = e ? (uint32_t)atoi(e) : 10u; return t < 8u ? 8u : (t > 11u ? 11u : t); }();
    return v;
}

template <class F>
__device__ __forceinline__ F load_el(const F *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 lo = q[0], hi = q[1];
    F r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
template <class F>
__device__ __forceinline__ void store_el(F *p, const F &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

__device__ __forceinline__ Fr29 lds_get(const int32_t *lds, uint32_t N, uint32_t e) {
    Fr29 r;
#pragma unroll
    for (int l = 0; l < 9; l++) r.l[l] = lds[l * N + e];
    return r;
}
__device__ __forceinline__ void lds_put(int32_t *lds, uint32_t N, uint32_t e, const Fr29 &r) {
#pragma unroll
    for (int l = 0; l < 9; l++) lds[l * N + e] = r.l[l];
}

// One pass over index bits [lo, lo+t): 2^t rows x 2^q contiguous columns per workgroup.
//   DIF : stages from bit lo+t-1 down to lo:  (u,v) -> (u+v, (u-v)*w)
//   DIT : stages from bit lo up to lo+t-1:    (u,v) -> (u+v*w, u-v*w)
// w = tw[j << (logn-1-b)] = w_n^(j*n/2^(b+1)), j = low b bits of the element index.
// premul (optional): element i is multiplied by premul[i] as it is loaded.
// Lazy ranges (field29.hpp): a DIT value grows by < 1.2p per stage (11 stages: < 15p, and only
// ever meets a canonical twiddle in a product); the DIF sum u+v doubles per stage, so it is
// pulled back to (-p/2, p/2) every third stage.  Stores canonicalise.
__device__ __forceinline__ Fr29 load_tw(const TwEntry *e) {
    const uint4 *q = reinterpret_cast<const uint4 *>(e);
    const uint4 a = q[0], b = q[1], c = q[2];
    Fr29 w;
    w.l[0] = (int32_t)a.x; w.l[1] = (int32_t)a.y; w.l[2] = (int32_t)a.z; w.l[3] = (int32_t)a.w;
    w.l[4] = (int32_t)b.x; w.l[5] = (int32_t)b.y; w.l[6] = (int32_t)b.z; w.l[7] = (int32_t)b.w;
    w.l[8] = (int32_t)c.x;
    return w;
}
__device__ __forceinline__ void store_tw(TwEntry *e, const Fr29 &v) {       // v canonical: limbs in [0, 2^29)
    uint4 *q = reinterpret_cast<uint4 *>(e);
    q[0] = make_uint4((uint32_t)v.l[0], (uint32_t)v.l[1], (uint32_t)v.l[2], (uint32_t)v.l[3]);
    q[1] = make_uint4((uint32_t)v.l[4], (uint32_t)v.l[5], (uint32_t)v.l[6], (uint32_t)v.l[7]);
    q[2] = make_uint4((uint32_t)v.l[8], 0u, 0u, 0u);
}

template <bool DIF, int NTT_THREADS>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(Fr *data, uint64_t stride_elems, const TwEntry *tw, const Fr *premul,
                                                          uint32_t logn, uint32_t lo, uint32_t t, uint32_t q) {
    extern __shared__ int32_t lds[];
    const uint32_t T = t + q, N = 1u << T;
    Fr *x = data + (uint64_t)blockIdx.y * stride_elems;
    const uint32_t tile = blockIdx.x;
    const uint32_t midw = lo - q;
    const uint64_t mid = tile & ((1u << midw) - 1u);
    const uint64_t hi = tile >> midw;
    const uint64_t base = (hi << (lo + t)) | (mid << q);
    const uint32_t cmask = (1u << q) - 1u;

    for (uint32_t e = threadIdx.x; e < N; e += NTT_THREADS) {
        uint64_t i = base | ((uint64_t)(e >> q) << lo) | (e & cmask);
        Fr29 v = Fr29::load(load_el(x + i));
        if (premul) v = Fr29::mul(v, Fr29::load(load_el(premul + i)));
        lds_put(lds, N, e, v);
    }
    __syncthreads();

    for (uint32_t s = 0; s < t; s++) {
        const uint32_t rb = DIF ? (t - 1 - s) : s;
        const uint32_t b = lo + rb;
        const uint32_t rmask = (1u << rb) - 1u;
        const uint32_t tshift = logn - 1 - b;
        for (uint32_t bt = threadIdx.x; bt < (N >> 1); bt += NTT_THREADS) {
            uint32_t c = bt & cmask, rp = bt >> q;
            uint32_t r0 = ((rp >> rb) << (rb + 1)) | (rp & rmask);
            uint32_t e0 = (r0 << q) | c;
            uint32_t e1 = e0 + (1u << (rb + q));
            uint64_t j = ((uint64_t)(r0 & rmask) << lo) | (mid << q) | c;
            Fr29 u = lds_get(lds, N, e0), v = lds_get(lds, N, e1);
            Fr29 s0, s1;
            if (b == 0) {                       // w = 1
                s0 = Fr29::add(u, v);
                s1 = Fr29::sub(u, v);
            } else {
                Fr29 w = load_tw(tw + (j << tshift));
                if (DIF) {
                    // the sum doubles per stage: pull it back every third stage (< 8p in between,
                    // well inside the (-13p, 13p) the 2^261 radix tolerates; the stores canonicalise)
                    s0 = Fr29::add(u, v);
                    if (s % 3u == 2u) s0 = Fr29::reduce_near_zero(s0);
                    s1 = Fr29::mul(Fr29::sub(u, v), w);
                } else {
                    v = Fr29::mul(v, w);
                    s0 = Fr29::add(u, v);
                    s1 = Fr29::sub(u, v);
                }
            }
            lds_put(lds, N, e0, s0);
            lds_put(lds, N, e1, s1);
        }
        __syncthreads();
    }

    for (uint32_t e = threadIdx.x; e < N; e += NTT_THREADS) {
        uint64_t i = base | ((uint64_t)(e >> q) << lo) | (e & cmask);
        store_el(x + i, Fr29::store(lds_get(lds, N, e)));
    }
}

struct PassPlan {
    uint32_t lo[8], t[8], q[8];
    int n;
};

// low pass covers bits [0, t0), t0 <= 11 (contiguous tile); the remaining bits are split
// evenly into passes of <= 8 rows-bits with q = min(11 - t, lo) column bits.
static PassPlan plan_passes(uint32_t logn) {
    PassPlan p;
    p.n = 0;
    const uint32_t TL = ntt_tile_log();
    uint32_t t0 = logn < TL ? logn : TL;
    p.lo[0] = 0; p.t[0] = t0; p.q[0] = 0; p.n = 1;
    uint32_t rem = logn - t0;
    if (rem) {
        uint32_t k = (rem + 7) / 8;
        uint32_t lo = t0;
        for (uint32_t i = 0; i < k; i++) {
            uint32_t t = rem / (k - i) + ((rem % (k - i)) ? 1 : 0);
            uint32_t q = TL - t;
            if (q > lo) q = lo;
            p.lo[p.n] = lo; p.t[p.n] = t; p.q[p.n] = q; p.n++;
            lo += t;
            rem -= t;
        }
    }
    return p;   // ascending bit order: DIT runs 0..n-1, DIF runs n-1..0
}

// local_logn: size of the array the pass runs over (== logn unless this is one block of a transform
// partitioned across GPUs: then the twiddles are still those of the full 2^logn domain)
template <bool DIF>
static void run_pass(Fr *data, uint64_t stride, uint32_t batch, const TwEntry *tw, const Fr *premul, uint32_t logn,
                     uint32_t lo, uint32_t t, uint32_t q, hipStream_t s, uint32_t local_logn) {
    uint32_t T = t + q;
    uint32_t tiles = 1u << (local_logn - T);
    size_t shmem = (size_t)36 << T;     // 9 limb planes; 72 KiB at T = 11 (opt-in above 64 KiB)
    static PerDeviceOnce attr;
    if (attr.need()) {
        ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_pass<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_pass<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_pass<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_pass<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done();
    }
    if (ntt_threads() == 512u)
        ZK_LAUNCH((k_ntt_pass<DIF, 512>), dim3(tiles, batch), dim3(512), shmem, s, data, stride, tw, premul, logn, lo, t, q);
    else
        ZK_LAUNCH((k_ntt_pass<DIF, 256>), dim3(tiles, batch), dim3(256), shmem, s, data, stride, tw, premul, logn, lo, t, q);
    ZK_LAUNCH_OK("ntt pass");
}

// Stages over index bits [0, local_logn) of a 2^logn transform on a contiguous block of 2^local_logn
// elements: the whole transform when local_logn == logn, one GPU's block of a partitioned one otherwise
// (the stages over the top logn - local_logn bits are launch_ntt_cross's).
void launch_ntt_dif_inverse(Fr *data, uint64_t stride, uint32_t batch, const NttTables &tb, hipStream_t s, uint32_t local_logn) {
    if (local_logn == NTT_FULL) local_logn = tb.logn;
    if (local_logn == 0) return;
    PassPlan p = plan_passes(local_logn);
    for (int i = p.n - 1; i >= 0; i--) run_pass<true>(data, stride, batch, tb.inv, nullptr, tb.logn, p.lo[i], p.t[i], p.q[i], s, local_logn);
}

void launch_ntt_dit_forward(Fr *data, uint64_t stride, uint32_t batch, const NttTables &tb, hipStream_t s, const Fr *premul, uint32_t local_logn) {
    if (local_logn == NTT_FULL) local_logn = tb.logn;
    if (local_logn == 0) {
        if (premul) launch_fr_scale_by_table(data, stride, batch, premul, 1, s);
        return;
    }
    PassPlan p = plan_passes(local_logn);
    for (int i = 0; i < p.n; i++)
        run_pass<false>(data, stride, batch, tb.fwd, i == 0 ? premul : nullptr, tb.logn, p.lo[i], p.t[i], p.q[i], s, local_logn);
}

// ------------------------------------------------------------------ transform partitioned across G = 2^g GPUs
// GPU r owns the contiguous block [r*n/G, (r+1)*n/G) of every polynomial (the same slices the H
// point table is sharded by), so the stages over the low logn-g index bits are local (above) and
// only the g stages over the TOP bits pair elements of different GPUs: element o of block s with
// element o of block s ^ 2^k.  Those g stages are one radix-G butterfly per block offset o.  GPU r
// does the butterflies of the offsets [r*n/G^2, (r+1)*n/G^2) for all G blocks: it receives that
// chunk of every block (all-to-all #1: RCCL all_to_all_single between processes, peer writes inside
// one process), runs k_ntt_cross on the G x n/G^2 exchange buffer `xb` (layout [poly][block][o']),
// and the results travel back to their blocks (peer writes by the kernel itself, or all-to-all #2).
// Traffic per GPU and transform: 2 x (G-1)/G of its block (28 MiB at 2^22, G = 8) — against the
// 112 MiB an all-gather of the blocks would move.  DIF (inverse transform) runs the cross stages
// FIRST, DIT (forward) runs them LAST; twiddles come from the full-size tables.
struct CrossOut {
    Fr *base[8];          // where the result for block s goes (peer memory or xb itself)
};

// stage T of the LG cross stages on the G = 2^LG values of one block offset (template recursion: the optimiser would not
// unroll the stage loop of the radix-8 case, which put the eight values into scratch memory)
template <bool DIF, int LG, int T>
__device__ __forceinline__ void cross_stage(Fr29 (&x)[1 << LG], const TwEntry *tw, uint32_t logn, uint64_t o, uint64_t block) {
    constexpr int G = 1 << LG;
    // DIF: bit b = logn-1-T, partner differs in block bit LG-1-T; DIT: b = logn-LG+T, block bit T
    constexpr int kb = DIF ? LG - 1 - T : T;
    const uint32_t b = DIF ? logn - 1 - T : logn - LG + T;
    const uint32_t tshift = logn - 1 - b;
#pragma unroll
    for (int pi = 0; pi < G / 2; pi++) {          // the G/2 butterflies of the stage: pi with a zero inserted at bit kb
        const int sidx = ((pi >> kb) << (kb + 1)) | (pi & ((1 << kb) - 1));
        const int s1 = sidx | (1 << kb);
        // j = low b bits of the global index s*block + o
        const uint64_t j = (uint64_t)(sidx & ((1 << kb) - 1)) * block + o;
        const Fr29 w = load_tw(tw + (j << tshift));
        Fr29 u = x[sidx], v = x[s1];
        if (DIF) {
            x[sidx] = Fr29::add(u, v);
            x[s1] = Fr29::mul(Fr29::sub(u, v), w);
        } else {
            v = Fr29::mul(v, w);
            x[sidx] = Fr29::add(u, v);
            x[s1] = Fr29::sub(u, v);
        }
    }
    if constexpr (T + 1 < LG) cross_stage<DIF, LG, T + 1>(x, tw, logn, o, block);
}

template <bool DIF, int LG>
__global__ __launch_bounds__(256) void k_ntt_cross(const Fr *xb, uint64_t in_src_stride, uint64_t in_poly_stride, CrossOut out,
                                                   uint64_t out_poly_stride, uint64_t out_offset,
                                                   const TwEntry *tw, uint32_t logn, uint32_t rank, uint64_t chunk, uint64_t block) {
    ZK_CHAIN_PRIO();
    constexpr int G = 1 << LG;
    const uint64_t op = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // offset inside this GPU's chunk
    if (op >= chunk) return;
    const uint32_t poly = blockIdx.y;
    const Fr *src = xb + (uint64_t)poly * in_poly_stride + op;
    const uint64_t o = (uint64_t)rank * chunk + op;                             // offset inside a block
    Fr29 x[G];
#pragma unroll
    for (int sidx = 0; sidx < G; sidx++) x[sidx] = Fr29::load(load_el(src + (uint64_t)sidx * in_src_stride));
    cross_stage<DIF, LG, 0>(x, tw, logn, o, block);
#pragma unroll
    for (int sidx = 0; sidx < G; sidx++)
        store_el(out.base[sidx] + (uint64_t)poly * out_poly_stride + out_offset + op, Fr29::store(x[sidx]));
}

void launch_ntt_cross(bool inverse, const Fr *xb, uint64_t in_src_stride, uint64_t in_poly_stride, Fr *const out_base[8],
                      uint64_t out_poly_stride, uint64_t out_offset, uint32_t batch,
                      const NttTables &tb, uint32_t log_shards, uint32_t rank, hipStream_t s) {
    if (log_shards == 0) return;
    const uint64_t block = 1ull << (tb.logn - log_shards), chunk = block >> log_shards;
    CrossOut out;
    for (int i = 0; i < 8; i++) out.base[i] = i < (1 << log_shards) ? out_base[i] : nullptr;
    const dim3 grid((uint32_t)((chunk + 255) / 256), batch), blk(256);
    const TwEntry *tw = inverse ? tb.inv : tb.fwd;
#define ZK_CROSS(D, L) ZK_LAUNCH((k_ntt_cross<D, L>), grid, blk, 0, s, xb, in_src_stride, in_poly_stride, out, out_poly_stride, out_offset, tw, tb.logn, rank, chunk, block)
    if (inverse) {
        if (log_shards == 1) ZK_CROSS(true, 1); else if (log_shards == 2) ZK_CROSS(true, 2); else ZK_CROSS(true, 3);
    } else {
        if (log_shards == 1) ZK_CROSS(false, 1); else if (log_shards == 2) ZK_CROSS(false, 2); else ZK_CROSS(false, 3);
    }
#undef ZK_CROSS
    ZK_LAUNCH_OK("ntt cross-GPU stages");
}

// all-to-all #1 inside one process: chunk s of every polynomial of this GPU's block -> slot `rank` of
// GPU s's exchange buffer (peer writes over xGMI; coalesced 32-byte elements)
struct ChunkDst {
    Fr *base[8];
};
// chunk c of polynomial `poly` of this GPU's block  <->  dst.base[c] + poly*poly_stride + offset (+ position in the chunk)
template <bool GATHER>
__global__ __launch_bounds__(256) void k_chunk_move(ChunkDst far, Fr *blockdata, uint64_t chunk, uint64_t block, uint64_t poly_stride, uint64_t offset) {
    ZK_CHAIN_PRIO();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // element inside this GPU's block
    if (i >= block) return;
    const uint32_t poly = blockIdx.y;
    const uint64_t sidx = i / chunk, op = i - sidx * chunk;
    Fr *there = far.base[sidx] + (uint64_t)poly * poly_stride + offset + op, *here = blockdata + (uint64_t)poly * block + i;
    if (GATHER) store_el(here, load_el(there));
    else store_el(there, load_el(here));
}
static void chunk_move(bool gather, Fr *const far_base[8], Fr *blockdata, uint64_t poly_stride, uint64_t offset, uint32_t batch, uint32_t logn,
                       uint32_t log_shards, hipStream_t s) {
    const uint64_t block = 1ull << (logn - log_shards), chunk = block >> log_shards;
    ChunkDst d;
    for (int i = 0; i < 8; i++) d.base[i] = i < (1 << log_shards) ? far_base[i] : nullptr;
    const dim3 grid((uint32_t)((block + 255) / 256), batch);
    if (gather) ZK_LAUNCH(k_chunk_move<true>, grid, dim3(256), 0, s, d, blockdata, chunk, block, poly_stride, offset);
    else ZK_LAUNCH(k_chunk_move<false>, grid, dim3(256), 0, s, d, blockdata, chunk, block, poly_stride, offset);
    ZK_LAUNCH_OK("chunk scatter/gather");
}
void launch_chunk_scatter(Fr *const xb_of_gpu[8], const Fr *blockdata, uint32_t batch, uint32_t logn, uint32_t log_shards, uint32_t rank,
                          hipStream_t s) {
    const uint64_t block = 1ull << (logn - log_shards), chunk = block >> log_shards;
    chunk_move(false, xb_of_gpu, const_cast<Fr *>(blockdata), block, (uint64_t)rank * chunk, batch, logn, log_shards, s);
}
// one-process-per-GPU path: the block <-> ONE contiguous buffer laid out [destination / source GPU][poly][chunk], so that
// a single all_to_all_single moves all three polynomials
void launch_chunk_pack(Fr *packed, const Fr *blockdata, uint32_t batch, uint32_t logn, uint32_t log_shards, hipStream_t s) {
    const uint64_t chunk = (1ull << (logn - log_shards)) >> log_shards;
    Fr *base[8];
    for (int i = 0; i < 8; i++) base[i] = packed + (uint64_t)i * batch * chunk;
    chunk_move(false, base, const_cast<Fr *>(blockdata), chunk, 0, batch, logn, log_shards, s);
}
void launch_chunk_unpack(Fr *blockdata, const Fr *packed, uint32_t batch, uint32_t logn, uint32_t log_shards, hipStream_t s) {
    const uint64_t chunk = (1ull << (logn - log_shards)) >> log_shards;
    Fr *base[8];
    for (int i = 0; i < 8; i++) base[i] = const_cast<Fr *>(packed) + (uint64_t)i * batch * chunk;
    chunk_move(true, base, blockdata, chunk, 0, batch, logn, log_shards, s);
}

// ------------------------------------------------------------------ pointwise helpers
__global__ __launch_bounds__(256) void k_scale_table(Fr *data, uint64_t stride_elems, const Fr *table, uint64_t n) {
    Fr *x = data + (uint64_t)blockIdx.y * stride_elems;
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st)
        store_el(x + i, Fr29::store(Fr29::mul(Fr29::load(load_el(x + i)), Fr29::load(load_el(table + i)))));
}
void launch_fr_scale_by_table(Fr *data, uint64_t stride, uint32_t batch, const Fr *table, uint64_t n, hipStream_t s) {
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    ZK_LAUNCH(k_scale_table, dim3((uint32_t)g, batch), dim3(256), 0, s, data, stride, table, n);
    ZK_LAUNCH_OK("scale by table");
}

__global__ __launch_bounds__(256) void k_scale_const(Fr *x, const Fr *k, uint64_t n) {
    Fr29 kk = Fr29::load(load_el(k));
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st)
        store_el(x + i, Fr29::store(Fr29::mul(Fr29::load(load_el(x + i)), kk)));
}
void launch_fr_scale_const(Fr *data, const Fr *k, uint64_t n, hipStream_t s) {
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    ZK_LAUNCH(k_scale_const, dim3((uint32_t)g), dim3(256), 0, s, data, k, n);
    ZK_LAUNCH_OK("scale");
}

__device__ __forceinline__ uint32_t brev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

__global__ __launch_bounds__(256) void k_bitrev(Fr *x, uint32_t logn) {
    uint64_t n = 1ull << logn;
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        uint32_t j = brev((uint32_t)i, logn);
        if (i < j) {
            Fr a = load_el(x + i), b = load_el(x + j);
            store_el(x + i, b);
            store_el(x + j, a);
        }
    }
}
void launch_bitrev_permute(Fr *data, uint32_t logn, hipStream_t s) {
    uint64_t n = 1ull << logn;
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    ZK_LAUNCH(k_bitrev, dim3((uint32_t)g), dim3(256), 0, s, data, logn);
    ZK_LAUNCH_OK("bit reversal");
}

// h[i] = fromMontgomery(a[i]*b[i] - c[i])  (src/groth16.cpp:158-163): standard-form MSM scalars
// blockIdx.y = vector of a batched submission (a|b|c at + y * abc_stride, h at + y * n)
__global__ __launch_bounds__(256) void k_abc_to_h(Fr *h, const Fr *a, const Fr *b, const Fr *c, uint64_t n, uint64_t abc_stride) {
    ZK_CHAIN_PRIO();
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    a += (uint64_t)blockIdx.y * abc_stride;
    b += (uint64_t)blockIdx.y * abc_stride;
    c += (uint64_t)blockIdx.y * abc_stride;
    h += (uint64_t)blockIdx.y * n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        Fr29 t = Fr29::sub(Fr29::mul(Fr29::load(load_el(a + i)), Fr29::load(load_el(b + i))), Fr29::load(load_el(c + i)));
        store_el(h + i, Fr29::store(Fr29::from_mont(t)));
    }
}
void launch_abc_to_h(Fr *h, const Fr *a, const Fr *b, const Fr *c, uint64_t n, hipStream_t s, uint32_t vectors, uint64_t abc_stride) {
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    ZK_LAUNCH(k_abc_to_h, dim3((uint32_t)g, vectors ? vectors : 1), dim3(256), 0, s, h, a, b, c, n, abc_stride);
    ZK_LAUNCH_OK("abc_to_h");
}

// x*2^256 (the reference's Montgomery form) <-> x*2^261 (this library's internal form), in place
__global__ __launch_bounds__(256) void k_fr_convert(Fr *x, uint64_t n, int to_internal, int times) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        Fr v = load_el(x + i);
        if (to_internal) {
            Fr29 t = Fr29::from_mont256(v);
            for (int k = 1; k < times; k++) t = Fr29::mul(t, Fr29::k_in());     // each extra round is another * 2^5
            store_el(x + i, Fr29::store(t));
        } else {
            store_el(x + i, Fr29::to_mont256(Fr29::load(v)));
        }
    }
}
void launch_fr_to_internal(Fr *x, uint64_t n, int times, hipStream_t s) {
    if (!n) return;
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    ZK_LAUNCH(k_fr_convert, dim3((uint32_t)g), dim3(256), 0, s, x, n, 1, times);
    ZK_LAUNCH_OK("fr_to_internal");
}
void launch_fr_from_internal(Fr *x, uint64_t n, hipStream_t s) {
    if (!n) return;
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    ZK_LAUNCH(k_fr_convert, dim3((uint32_t)g), dim3(256), 0, s, x, n, 0, 1);
    ZK_LAUNCH_OK("fr_from_internal");
}

// ------------------------------------------------------------------ twiddle tables
// w_{2^28} = 5^((r-1)/2^28) (snarkjs/ffjavascript convention), standard form:
// 19103219067921713944291392827692070036145651957329286315305642004821462161904
__device__ __constant__ const uint32_t ROOT_2_28_STD[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu,
                                                           0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};

__device__ Fr fr_pow(Fr base, uint64_t e) {
    Fr r = Fr::one();
    while (e) {
        if (e & 1) r = Fr::mul(r, base);
        base = Fr::sqr(base);
        e >>= 1;
    }
    return r;
}

// w_{2^k} in Montgomery form
__device__ Fr fr_root_of_unity(uint32_t k) {
    Fr w;
#pragma unroll
    for (int i = 0; i < 8; i++) w.v[i] = ROOT_2_28_STD[i];
    w = Fr::to_mont(w);
    for (uint32_t i = k; i < 28; i++) w = Fr::sqr(w);
    return w;
}

__global__ __launch_bounds__(256) void k_build_tables(TwEntry *fwd, TwEntry *inv, Fr *coset, Fr *ninv_out, uint32_t logn) {
    const uint64_t n = 1ull << logn;
    __shared__ Fr s_wn, s_wninv, s_w2n, s_ninv;
    if (threadIdx.x == 0) {
        Fr wn = fr_root_of_unity(logn);
        s_wn = wn;
        s_wninv = Fr::inv(wn);
        // the coset needs a root of order 2n: none exists for n = 2^28 (2-adicity 28) — the table is then
        // zero and never used (zk_prover_create and zk_fr_abc_to_h refuse such domains)
        s_w2n = logn < 28 ? fr_root_of_unity(logn + 1) : Fr::zero();
        // n^-1: Montgomery form of n is to_mont(n)
        Fr nn = Fr::zero();
        nn.v[0] = (uint32_t)n;
        nn.v[1] = (uint32_t)(n >> 32);
        s_ninv = Fr::inv(Fr::to_mont(nn));
    }
    __syncthreads();
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        if (i < (n >> 1) || n == 1) {
            if (n > 1) {
                store_tw(fwd + i, Fr29::canonical(Fr29::from_mont256(fr_pow(s_wn, i))));
                store_tw(inv + i, Fr29::canonical(Fr29::from_mont256(fr_pow(s_wninv, i))));
            }
        }
        uint32_t k = brev((uint32_t)i, logn);
        if (coset) store_el(coset + i, Fr29::store(Fr29::from_mont256(Fr::mul(fr_pow(s_w2n, k), s_ninv))));
        if (i == 0) store_el(ninv_out, Fr29::store(Fr29::from_mont256(s_ninv)));
    }
}

void launch_ntt_build_tables(TwEntry *fwd, TwEntry *inv, Fr *coset, Fr *ninv, uint32_t logn, hipStream_t s) {
    uint64_t n = 1ull << logn;
    uint64_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    ZK_LAUNCH(k_build_tables, dim3((uint32_t)g), dim3(256), 0, s, fwd, inv, coset, ninv, logn);
    ZK_LAUNCH_OK("twiddle tables");
}

}   // namespace zk
