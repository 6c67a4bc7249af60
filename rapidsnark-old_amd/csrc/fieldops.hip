// Pointwise Fr/Fq kernels and the sparse A.w / B.w accumulation (src/groth16.cpp:56-96).
#include "common.hpp"
#include "kernels.hpp"
#include "hipcheck.hpp"
#include "field29.hpp"

namespace zk {

// 32-byte elements move as two 16-byte (dwordx4) accesses per lane: fully coalesced.
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

template <class F>
__global__ __launch_bounds__(256) void k_mul_vec(F *out, const F *a, const F *b, uint64_t n) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        store_el(out + i, F::mul(load_el(a + i), load_el(b + i)));
}

static inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap = 256 * 8) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (uint32_t)(g > cap ? cap : g);
}

void launch_fr_mul_vec(Fr *out, const Fr *a, const Fr *b, uint64_t n, hipStream_t s) {
    ZK_LAUNCH(k_mul_vec<Fr>, dim3(grid_for(n, 256)), dim3(256), 0, s, out, a, b, n);
    ZK_LAUNCH_OK("fr_mul_vec");
}
void launch_fq_mul_vec(Fq *out, const Fq *a, const Fq *b, uint64_t n, hipStream_t s) {
    ZK_LAUNCH(k_mul_vec<Fq>, dim3(grid_for(n, 256)), dim3(256), 0, s, out, a, b, n);
    ZK_LAUNCH_OK("fq_mul_vec");
}

// One lane per domain row i: a[i] = sum_A coef*w[s], b[i] = sum_B coef*w[s], c[i] = a[i]*b[i].
// The reference does this with 1024 striped omp locks (src/groth16.cpp:63-84); a row-sorted
// CSR built once at create time needs neither locks nor atomics.  The zkey stores
// value*2^512; create rescales it to value*2^522 so that one 2^-261 Montgomery product with
// the standard-form witness gives w*value in this library's 2^261 form (field29.hpp).
// blockIdx.y = vector of a batched submission: its witness at wtns + y * wtns_stride, its a|b|c at + y * abc_stride
__global__ __launch_bounds__(256) void k_spmv_abc(Fr *a, Fr *b, Fr *c, CsrDev csr, const Fr *wtns, uint32_t n, uint64_t abc_stride, uint64_t wtns_stride) {
    ZK_CHAIN_PRIO();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a += (uint64_t)blockIdx.y * abc_stride;
    b += (uint64_t)blockIdx.y * abc_stride;
    c += (uint64_t)blockIdx.y * abc_stride;
    wtns += (uint64_t)blockIdx.y * wtns_stride;
    Fr29 acc[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        uint32_t row = i + (m ? n : 0);
        uint32_t lo = csr.rowptr[row], hi = csr.rowptr[row + 1];
        Fr29 sum = Fr29::zero();
        uint32_t pending = 0;
        for (uint32_t k = lo; k < hi; k++) {
            Fr29 w = Fr29::load(load_el(wtns + csr.col[k]));       // standard form, < r for well-formed files
            Fr29 v = Fr29::load(load_el(csr.val + k));             // value * 2^522 (pre-scaled at create)
            sum = Fr29::add(sum, Fr29::mul(w, v));                 // += w*value * 2^261
            if (++pending == 8) {                                  // long rows: keep the lazy sum small
                sum = Fr29::reduce_near_zero(sum);
                pending = 0;
            }
        }
        acc[m] = Fr29::reduce_near_zero(sum);
    }
    store_el(a + i, Fr29::store(acc[0]));
    store_el(b + i, Fr29::store(acc[1]));
    store_el(c + i, Fr29::store(Fr29::mul(acc[0], acc[1])));
}

void launch_spmv_abc(Fr *a, Fr *b, Fr *c, CsrDev csr, const Fr *wtns, uint32_t n, hipStream_t s, uint32_t vectors, uint64_t abc_stride, uint64_t wtns_stride) {
    ZK_LAUNCH(k_spmv_abc, dim3((n + 255) / 256, vectors ? vectors : 1), dim3(256), 0, s, a, b, c, csr, wtns, n, abc_stride, wtns_stride);
    ZK_LAUNCH_OK("spmv_abc");
}

// ---- CSR build on the device (the reference has no such step: it walks the records under 1024
// striped locks on every proof, src/groth16.cpp:63-84).  Records are 44-byte packed, 4-byte aligned.
// Rows outside [row_lo, row_hi) are skipped (a prover that holds one block of a partitioned chain keeps
// only its rows); the range check covers every record either way.  Local row = c - row_lo, rows of
// matrix B follow the nl = row_hi - row_lo rows of matrix A.
__global__ __launch_bounds__(256) void k_csr_count(uint32_t *rowcount, uint32_t *err, const uint32_t *rec, uint64_t nCoefs, uint32_t n,
                                                   uint32_t nVars, uint32_t row_lo, uint32_t row_hi) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t nl = row_hi - row_lo;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nCoefs; i += st) {
        const uint32_t *r = rec + i * 11;
        const uint32_t m = r[0] ? 1u : 0u, c = r[1], sg = r[2];      // the reference: (coefs[i].m == 0) ? a : b  (groth16.cpp:69)
        if (c >= n || sg >= nVars) {
            atomicOr(err, 1u);
            continue;
        }
        if (c < row_lo || c >= row_hi) continue;
        atomicAdd(&rowcount[(uint64_t)m * nl + (c - row_lo)], 1u);
    }
}
// The order of a row's terms depends on atomic arbitration; their (exact, modular) sum does not.
__global__ __launch_bounds__(256) void k_csr_fill(uint32_t *col, Fr *val, uint32_t *cursor, const uint32_t *rec, uint64_t nCoefs, uint32_t n,
                                                  uint32_t nVars, uint32_t row_lo, uint32_t row_hi) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t nl = row_hi - row_lo;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nCoefs; i += st) {
        const uint32_t *r = rec + i * 11;
        const uint32_t m = r[0] ? 1u : 0u, c = r[1], sg = r[2];
        if (c >= n || sg >= nVars || c < row_lo || c >= row_hi) continue;
        uint32_t pos = atomicAdd(&cursor[(uint64_t)m * nl + (c - row_lo)], 1u);
        col[pos] = sg;
        Fr v;
#pragma unroll
        for (int j = 0; j < 8; j++) v.v[j] = r[3 + j];
        store_el(val + pos, v);
    }
}

void launch_csr_build(uint32_t *rowptr, uint32_t *col, Fr *val, uint32_t *cursor, uint32_t *err, const uint8_t *records,
                      uint64_t nCoefs, uint32_t n, uint32_t nVars, uint32_t row_lo, uint32_t row_hi, hipStream_t s) {
    const uint32_t rows = 2 * (row_hi - row_lo);
    ZK_HIP(hipMemsetAsync(cursor, 0, (size_t)rows * 4, s));
    ZK_HIP(hipMemsetAsync(err, 0, 4, s));
    const uint32_t g = grid_for(nCoefs ? nCoefs : 1, 256, 256 * 16);
    ZK_LAUNCH(k_csr_count, dim3(g), dim3(256), 0, s, cursor, err, (const uint32_t *)records, nCoefs, n, nVars, row_lo, row_hi);
    launch_exclusive_scan_u32(rowptr, cursor, rows, s);
    ZK_HIP(hipMemcpyAsync(cursor, rowptr, (size_t)rows * 4, hipMemcpyDeviceToDevice, s));
    ZK_LAUNCH(k_csr_fill, dim3(g), dim3(256), 0, s, col, val, cursor, (const uint32_t *)records, nCoefs, n, nVars, row_lo, row_hi);
    ZK_LAUNCH_OK("csr build");
}

}   // namespace zk
