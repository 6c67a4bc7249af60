// Operator-level entry points of the C-ABI (host pointers, staged through the device): the counterparts of the reference's
// E.fr.mul / FFT::fft / Curve::multiMulByScalar call sites (src/groth16.cpp:91-95, 102-152, 171-204) as stand-alone
// operators — what the KATs, zkgen and the parity tests call — and the synthetic-table helpers of the benchmark.
#include "prover_internal.hpp"

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
