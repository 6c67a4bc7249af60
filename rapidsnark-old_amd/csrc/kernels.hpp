// Launchers of the gfx950 kernels (defined in ntt.hip, msm.hip, fieldops.hip).
// All pointers are device pointers; all launches are asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.hpp"
#include "curve.hpp"

namespace zk {

// ---------------------------------------------------------------- fieldops.hip
void launch_fr_mul_vec(Fr *out, const Fr *a, const Fr *b, uint64_t n, hipStream_t s);
void launch_fq_mul_vec(Fq *out, const Fq *a, const Fq *b, uint64_t n, hipStream_t s);

// Sparse A.w / B.w accumulation + c = a o b  (src/groth16.cpp:56-96).
// CSR over 2n rows: rows [0,n) are matrix A, rows [n,2n) are matrix B.
struct CsrDev {
    const uint32_t *rowptr;   // 2n+1
    const uint32_t *col;      // nnz: signal index s
    const Fr *val;            // nnz: coefficient (value*R^2 as stored in the zkey)
};
void launch_spmv_abc(Fr *a, Fr *b, Fr *c, CsrDev csr, const Fr *wtns, uint32_t n, hipStream_t s, uint32_t vectors = 1, uint64_t abc_stride = 0,
                     uint64_t wtns_stride = 0);     // vectors > 1: a batched submission, vector v at + v * stride
// Row-sorted CSR from the zkey's coefficient records (section 4 after its u32 count: 44-byte packed
// {u32 matrix, u32 row, u32 signal, 32-byte value}, src/groth16.hpp:27-35), built on the device.
// rowptr: 2n + 1 words + msm_scan_extra_words(2n) of scan scratch; cursor: 2n words of scratch;
// err: one word, set non-zero when a record is out of range (matrix > 1, row >= n, signal >= nVars).
// Only the rows [row_lo, row_hi) of both matrices are kept (local row = row - row_lo; rowptr/cursor then
// hold 2 * (row_hi - row_lo) rows); the range check covers every record.
void launch_csr_build(uint32_t *rowptr, uint32_t *col, Fr *val, uint32_t *cursor, uint32_t *err, const uint8_t *records,
                      uint64_t nCoefs, uint32_t n, uint32_t nVars, uint32_t row_lo, uint32_t row_hi, hipStream_t s);
// exclusive scan: out[i] = sum counts[0..i), out[total] = grand total; out holds total + 1 + msm_scan_extra_words(total) words
void launch_exclusive_scan_u32(uint32_t *out, const uint32_t *counts, uint32_t total, hipStream_t s);

// ---------------------------------------------------------------- ntt.hip
// A butterfly twiddle as the NTT kernel consumes it: the nine 29-bit limbs of the canonical value
// (2^261 Montgomery form), padded to 48 bytes so that an entry is three aligned 16-byte loads and
// needs no word->limb conversion per butterfly.
struct TwEntry {
    uint32_t l[12];
};
struct NttTables {
    uint32_t logn;       // domain size 2^logn
    const TwEntry *fwd;  // w_n^k,  k < n/2
    const TwEntry *inv;  // w_n^-k, k < n/2
    const Fr *coset;     // n^-1 * w_2n^brev(p) for position p < n  (bit-reversed order; canonical words of the 2^261 form)
    const Fr *ninv;      // single element n^-1 (Montgomery)
};
// ---------------------------------------------------------------- nttpair.hip
// The coset-evaluation pipeline of the proof path (ifft, coset shift, fft of a|b|c: src/groth16.cpp:98-155) with radix-8
// register butterflies; tables for ONE local transform size (the whole domain, or one block of a partitioned chain).
struct NttPair {
    uint32_t L = 0, Lg = 0, m = 0;        // local bits (0 = not built), domain bits, tile bits
    uint32_t ngroups = 0, g[2] = {0, 0};  // strided bits: none, one pass, or two
    TwEntry *rfwd = nullptr, *rinv = nullptr;
    Fr *tinv = nullptr, *tfwd = nullptr, *dtab = nullptr, *t2inv = nullptr, *t2fwd = nullptr;
    NttPair() {}
    NttPair(const NttPair &) = delete;
    NttPair &operator=(const NttPair &) = delete;
    ~NttPair() { release(); }
    // plain: the tables of the stand-alone transforms (launch_ntt_plain) instead of the coset-evaluation pair's
    void build(uint32_t logn_global, uint32_t logn_local, uint32_t block_index, hipStream_t s, bool plain = false);
    void release();
};
bool ntt_pair_supported(uint32_t local_logn);
// in place on `batch` vectors of 2^L elements, `stride_elems` apart: natural order in, natural order out
void launch_ntt_coset_pair(Fr *data, uint64_t stride_elems, uint32_t batch, const NttPair &t, hipStream_t s);
// one inverse (1/n included) or forward transform per vector on the same passes; `t` built with plain = true.  Input in
// `data` (destroyed), result in `out` (another buffer of the same size: the bit reversal is folded into the middle pass's
// addressing, which crosses workgroups).
void launch_ntt_plain(Fr *out, Fr *data, uint64_t stride_elems, uint32_t batch, const NttPair &t, bool inverse, hipStream_t s);

// fill the tables (device memory already allocated: n/2, n/2, n, 1 elements)
void launch_ntt_build_tables(TwEntry *fwd, TwEntry *inv, Fr *coset, Fr *ninv, uint32_t logn, hipStream_t s);
// Batched in-place transforms of `batch` polynomials laid out at data + k*stride_elems.
//  dif_inverse: natural -> bit-reversed, inverse twiddles, NO scaling
//  dit_forward: bit-reversed -> natural, forward twiddles
// local_logn < t.logn: only the stages over index bits [0, local_logn), on one contiguous block of
// 2^local_logn elements of a transform partitioned across GPUs (twiddles of the full domain)
#define NTT_FULL 0xffffffffu
void launch_ntt_dif_inverse(Fr *data, uint64_t stride_elems, uint32_t batch, const NttTables &t, hipStream_t s, uint32_t local_logn = NTT_FULL);
// premul (optional): table multiplied in as the first pass loads (the fused coset*1/n shift)
void launch_ntt_dit_forward(Fr *data, uint64_t stride_elems, uint32_t batch, const NttTables &t, hipStream_t s, const Fr *premul = nullptr,
                            uint32_t local_logn = NTT_FULL);
// The stages over the TOP log_shards index bits of a transform partitioned across G = 2^log_shards GPUs
// (see ntt.hip).  xb = this GPU's exchange buffer: element o' of polynomial `poly` of source GPU s sits at
// xb[s*in_src_stride + poly*in_poly_stride + o'] ([poly][G][n/G^2] inside one process, [G][poly][n/G^2] on the
// all_to_all path); the result for block s is written to out_base[s] + poly*out_poly_stride + out_offset + o'.
// inverse: DIF stages + inverse twiddles.
void launch_ntt_cross(bool inverse, const Fr *xb, uint64_t in_src_stride, uint64_t in_poly_stride, Fr *const out_base[8],
                      uint64_t out_poly_stride, uint64_t out_offset, uint32_t batch,
                      const NttTables &t, uint32_t log_shards, uint32_t rank, hipStream_t s);
// chunk s of every polynomial of this GPU's block (batch x n/G elements) -> slot `rank` of xb_of_gpu[s]
void launch_chunk_scatter(Fr *const xb_of_gpu[8], const Fr *blockdata, uint32_t batch, uint32_t logn, uint32_t log_shards, uint32_t rank,
                          hipStream_t s);
// block <-> one contiguous buffer [GPU][poly][n/G^2] (what ONE all_to_all_single sends / delivers)
void launch_chunk_pack(Fr *packed, const Fr *blockdata, uint32_t batch, uint32_t logn, uint32_t log_shards, hipStream_t s);
void launch_chunk_unpack(Fr *blockdata, const Fr *packed, uint32_t batch, uint32_t logn, uint32_t log_shards, hipStream_t s);
// Fr vectors: x*2^256 (zkey/ffiasm Montgomery form) -> x*2^(256+5*times) ; and 2^261 -> 2^256.
// The NTT / SpMV kernels work on canonical words of the 2^261 form (field29.hpp).
void launch_fr_to_internal(Fr *x, uint64_t n, int times, hipStream_t s);
void launch_fr_from_internal(Fr *x, uint64_t n, hipStream_t s);
// x[p] *= table[p]  (coset shift fused with 1/n, src/groth16.cpp:107-110)
void launch_fr_scale_by_table(Fr *data, uint64_t stride_elems, uint32_t batch, const Fr *table, uint64_t n, hipStream_t s);
// x[p] *= k ; and natural<->bit-reversed permutation (operator-level zk_fr_ntt only)
void launch_fr_scale_const(Fr *data, const Fr *k, uint64_t n, hipStream_t s);
void launch_bitrev_permute(Fr *data, uint32_t logn, hipStream_t s);
// h[i] = fromMontgomery(a[i]*b[i] - c[i])  (src/groth16.cpp:158-163)
void launch_abc_to_h(Fr *h, const Fr *a, const Fr *b, const Fr *c, uint64_t n, hipStream_t s, uint32_t vectors = 1, uint64_t abc_stride = 0);

// ---------------------------------------------------------------- msm.hip
struct MsmPlan {
    uint32_t c;          // window bits
    uint32_t W;          // digit windows = ceil(256 / c)
    uint32_t nbuckets;   // buckets per bucket set = 2^(c-1)
    uint32_t sets;       // bucket sets: W normally, 1 with window-precomputed tables (2 with rows for every second window)
    uint32_t precomp;    // 1: tables hold 2^(c*j) P_i for j < W and every window shares one bucket set;
                         // 2: rows for the EVEN windows only (row u = 2^(2c*u) P_i, ceil(W/2) rows): window 2u + 1 uses row u too and
                         //    adds into a second bucket set, whose sum the host doubles c times (Horner over the two sets)
    uint32_t batch;      // >= 1: the scalar vector is `batch` vectors of batch_n scalars back to back (several small proofs
    uint32_t batch_n;    //   in one set of launches): vector v has its own bucket set v (sets == batch), every vector
};                       //   indexes the SAME table rows.  Window-precomputed tables only.
// n = scalars per vector; batch > 1 => the sort runs over batch * n scalars
MsmPlan make_msm_plan(uint64_t n, uint32_t window_bits, uint32_t precomp = 0, uint32_t batch = 1);
// rows of n points a table of this plan holds (1 = the points themselves)
inline uint32_t msm_table_rows(const MsmPlan &p) { return p.precomp ? (p.W + p.precomp - 1) / p.precomp : 1u; }

// Digit recoding + counting sort of a scalar vector into bucket order (shared by every MSM over
// that vector).  Buffer sizes (in elements) come from msm_sort_sizes(); unused ones are 0.
struct MsmSortSizes {
    uint64_t counts_u32, starts_u32, offsets_u32, entries_u32;
    uint64_t codes_u32, lo_u16, val_u32, bin_counts_u32, bin_starts_u32;
};
struct MsmSortBufs {
    uint32_t *offsets, *entries;          // outputs: bucket starts [sets*nbuckets + 1]; idx | sign<<31 in bucket order
    uint32_t *counts, *starts;            // per-(bucket, slice) counts and their exclusive scan
    uint32_t *codes, *val, *bin_counts, *bin_starts;   // 32-bit digit codes (window-major); bin-partitioned items
    uint16_t *lo;                         // low key bits of the bin-partitioned items
};
MsmSortSizes msm_sort_sizes(uint64_t n, MsmPlan p);
uint32_t msm_scan_extra_words(uint32_t total);
void launch_msm_sort(const MsmSortBufs &b, const Fr *scalars, uint64_t n, MsmPlan p, hipStream_t s);
// table: W*n affine points, rows [0, n) already hold P_i (internal form); fills rows [n, W*n).
// tmp: (W-1)*n XYZZ, pref: (W-1)*n field elements (scratch).
void launch_msm_precomp_g1(G1Affine *table, G1XYZZ *tmp, Fq *pref, uint64_t n, MsmPlan p, hipStream_t s);
void launch_msm_precomp_g2(G2Affine *table, G2XYZZ *tmp, Fq2 *pref, uint64_t n, MsmPlan p, hipStream_t s);
// Bucket sums, partial sums and reduction scratch in HBM: the accumulators' own nine 29-bit limbs per
// coordinate (lazy, not canonical — msm.hip), all-zero = infinity.  G1: x | y | zz | zzz; G2: the four
// real components, then the four imaginary ones (what a lane of a lane pair loads in one go).
struct alignas(16) G1Acc {
    int32_t l[36];
};
struct alignas(16) G2Acc {
    int32_t l[72];
};
// buckets[b] = sum of +-points[idx - idx_sub] over the entries of b with idx >= idx_min.
// max_entries: upper bound of offsets[total_buckets] (= n*W); ws_*: msm_accum_workspace_slots() slots.
// ev (optional): two events recorded immediately before/after the level-1 kernel.
uint64_t msm_accum_workspace_slots(uint64_t max_entries);
// Optional follow-up stream of an accumulation: the level-1 kernel runs on `s`, the (small,
// latency-bound) partial merges on `stream` once `l1_done` fires.
struct AccumTail {
    hipStream_t stream = nullptr;
    hipEvent_t l1_done = nullptr;
    bool buckets_zeroed = false;      // the caller has already cleared the bucket array (ordered before this launch)
    uint32_t chunk_min = 0;           // entries per level-1 lane at least (0: ACC_CHUNK_MIN); more = fewer lanes, fewer partial sums to merge
    uint32_t chunk_max = 0;           // entries per lane beyond which another round of lanes is launched (0: ACC_CHUNK_MAX)
};
// Up to three G1 MSMs over the SAME sorted entry list in one set of launches (blockIdx.y): MSM m uses
// points[m], writes buckets + m*bucket_stride and the workspaces + m*ws_stride.
struct AccumBatch {
    uint32_t n;
    const void *points[3];
    uint32_t idx_min[3], idx_sub[3];
    uint64_t bucket_stride, ws_stride;
    uint32_t gather_mask;          // 0xffffffff; anything else is a measurement probe (results are wrong)
};
void launch_msm_accum_g1_batch(G1Acc *buckets, const uint32_t *offsets, const uint32_t *entries, const AccumBatch &batch, uint32_t total_buckets,
                               uint64_t max_entries, G1Acc *ws_part, uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev = nullptr,
                               AccumTail tail = AccumTail());
void launch_msm_accum_g1(G1Acc *buckets, const uint32_t *offsets, const uint32_t *entries, const G1Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total_buckets, uint64_t max_entries,
                         G1Acc *ws_part, uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev = nullptr,
                         AccumTail tail = AccumTail());
void launch_msm_accum_g2(G2Acc *buckets, const uint32_t *offsets, const uint32_t *entries, const G2Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total_buckets, uint64_t max_entries,
                         G2Acc *ws_part, uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev = nullptr,
                         AccumTail tail = AccumTail());
// window_sums[m*W + w] = sum_k (k+1) * buckets[m][w][k]  for n_msm bucket arrays laid back to back;
// scratch: n_msm * W * nbuckets/REDUCE_CHUNK points
void launch_msm_reduce_g1(G1XYZZ *window_sums, G1Acc *scratch, const G1Acc *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s);
void launch_msm_reduce_g2(G2XYZZ *window_sums, G2Acc *scratch, const G2Acc *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s);
uint64_t msm_reduce_scratch_points(uint32_t n_msm, MsmPlan p);
// window-sum records per bucket set the reduction writes: 1 (the set's sum) or, for small sets, c (T, S_0 .. S_{c-2}:
// set sum = T + sum_j 2^j S_j, finished on the host); layout [msm][set][record]
uint32_t msm_wsum_rc(MsmPlan p);

// MSM tables live in HBM as canonical words of x*2^261 (the 29-bit-limb kernels' Montgomery radix);
// converts n coordinates in place from the zkey's x*2^256.
void launch_fq_to_internal(Fq *coords, uint64_t n, hipStream_t s);

// ---------------------------------------------------------------- synth.hip
void launch_chain_g1(G1Affine *d_out, G1XYZZ *d_tmp, Fq *d_pref, const G1Affine &P0, const G1Affine &Q, uint64_t n, hipStream_t s);
void launch_chain_g2(G2Affine *d_out, G2XYZZ *d_tmp, Fq2 *d_pref, const G2Affine &P0, const G2Affine &Q, uint64_t n, hipStream_t s);
// out[i] = scalars[i] * B  (scalars: n x 8 little-endian words, standard form)
void launch_fixed_base_g1(G1Affine *d_out, G1XYZZ *d_tmp, Fq *d_pref, const G1Affine &B, const uint32_t *d_scalars, uint64_t n, hipStream_t s);
void launch_fixed_base_g2(G2Affine *d_out, G2XYZZ *d_tmp, Fq2 *d_pref, const G2Affine &B, const uint32_t *d_scalars, uint64_t n, hipStream_t s);

}   // namespace zk
