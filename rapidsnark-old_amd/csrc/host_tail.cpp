// Host tail of a proof: window Horner combine, shard-sum, final assembly, JSON.
// Compiled by g++ (no HIP).  Mirrors src/groth16.cpp:209-301 and src/main_prover.cpp:77-93.
#include "common.hpp"
#include "field64.hpp"
#include "curve.hpp"
#include "../../include/zkhip.h"

#include <string.h>
#include <stdio.h>
#include <sys/random.h>
#include <list>
#include <memory>
#include <new>
#include <mutex>
#include <thread>
#include <vector>

namespace zk {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }

typedef Fp2T<Fq64> Fq2h;
typedef Affine<Fq64> G1A;
typedef Affine<Fq2h> G2A;
typedef XYZZ<Fq64> G1P;
typedef XYZZ<Fq2h> G2P;

static_assert(sizeof(G1A) == 64 && sizeof(G2A) == 128 && sizeof(G1P) == 128 && sizeof(G2P) == 256, "layout");

// One bucket set's sum from its rc records: rc = 1: the sum itself; rc = c: T, S_0 .. S_{c-2} of the bit-sum
// reduction (msm.hip): sum_k (k+1) B_k = T + sum_j 2^j S_j — a serial chain of c-1 doublings, microseconds here.
template <class PT>
static PT set_sum(const uint8_t *w, uint32_t rc) {
    PT acc = PT::inf(), s;
    for (uint32_t j = rc - 1; j >= 1; j--) {
        if (!acc.is_inf()) acc = dbl(acc);
        memcpy(&s, w + (size_t)j * sizeof(PT), sizeof(PT));
        add(acc, s);
    }
    memcpy(&s, w, sizeof(PT));
    add(acc, s);
    return acc;
}
template <class PT>
static PT horner(const uint8_t *w, uint32_t W, uint32_t c, uint32_t rc) {
    PT acc = PT::inf();
    for (int i = (int)W - 1; i >= 0; i--) {
        if (!acc.is_inf()) for (uint32_t k = 0; k < c; k++) acc = dbl(acc);
        add(acc, set_sum<PT>(w + (size_t)i * rc * sizeof(PT), rc));
    }
    return acc;
}
template <class PT, class AT>
static void combine_windows(const uint8_t *w, uint32_t W, uint32_t c, uint32_t rc, uint8_t *out) {
    AT a = to_affine(horner<PT>(w, W, c, rc));
    memcpy(out, &a, sizeof(AT));
}
void HostTail::combine_windows_g1(const uint8_t *w, uint32_t W, uint32_t c, uint32_t rc, uint8_t out[64]) {
    combine_windows<G1P, G1A>(w, W, c, rc, out);
}
void HostTail::combine_windows_g2(const uint8_t *w, uint32_t W, uint32_t c, uint32_t rc, uint8_t out[128]) {
    combine_windows<G2P, G2A>(w, W, c, rc, out);
}

template <class PT, class AT>
static void add_affine(uint8_t *acc, const uint8_t *in) {
    AT a, b;
    memcpy(&a, acc, sizeof(AT));
    memcpy(&b, in, sizeof(AT));
    PT p = PT::from_affine(a);
    madd(p, b);
    AT r = to_affine(p);
    memcpy(acc, &r, sizeof(AT));
}
void HostTail::add_affine_g1(uint8_t acc[64], const uint8_t in[64]) { add_affine<G1P, G1A>(acc, in); }
void HostTail::add_affine_g2(uint8_t acc[128], const uint8_t in[128]) { add_affine<G2P, G2A>(acc, in); }

template <class AT>
static AT load(const uint8_t *b) {
    AT a;
    memcpy(&a, b, sizeof(AT));
    return a;
}

// ---- scalar multiplications of the final assembly -------------------------------------------------
// The reference's tail (groth16.cpp:222-246) is six 256-bit double-and-add multiplications; done that
// way they cost 0.84 ms of host time per proof here — more than the whole GPU side of a 2^14 circuit.
// Four of the six multiply the key's delta (r*delta1, s*delta1, rs*delta1, s*delta2): fixed bases, so a
// table of m * 16^j * delta (m = 1..8, j = 0..64, affine) turns each into <= 65 mixed additions.  The
// other two (s*pi_a, r*pi_b1) are summed anyway: one joint signed-window pass shares their doublings.

// k = sum_j d[j] * 16^j with d[j] in [-8, 7] (j < 64) and d[64] in {0, 1}
static void recode16(const u32 k[8], int8_t d[65]) {
    int carry = 0;
    for (int j = 0; j < 64; j++) {
        int v = (int)((k[j >> 3] >> ((j & 7) * 4)) & 15u) + carry;
        carry = v >= 8;
        d[j] = (int8_t)(carry ? v - 16 : v);
    }
    d[64] = (int8_t)carry;
}

template <class F>
static Affine<F> neg_affine(const Affine<F> &a) { return Affine<F>{a.x, F::neg(a.y)}; }

template <class F>
struct FixedBase {
    typedef XYZZ<F> PT;
    typedef Affine<F> AT;
    std::vector<AT> t;          // t[j*8 + m-1] = m * 16^j * base; empty = no table (use the generic multiplication)
    bool build(const AT &base) {
        if (base.is_inf()) return false;
        std::vector<PT> proj(65 * 8);
        PT b = PT::from_affine(base);
        for (int j = 0; j < 65; j++) {
            PT m = b;
            proj[j * 8] = m;
            for (int i = 1; i < 8; i++) { add(m, b); proj[j * 8 + i] = m; }
            b = dbl(m);                                    // 16 * (16^j base) = 2 * (8 * 16^j base)
        }
        // one inversion for all of them: prefix products of u_i = zz_i * zzz_i
        const size_t n = proj.size();
        std::vector<F> u(n), pre(n);
        F run = F::one();
        for (size_t i = 0; i < n; i++) {
            if (proj[i].is_inf()) return false;            // base of small order: not a key this table is for
            u[i] = F::mul(proj[i].zz, proj[i].zzz);
            pre[i] = run;
            run = F::mul(run, u[i]);
        }
        if (run.is_zero()) return false;
        F inv = F::inv(run);
        t.resize(n);
        for (size_t i = n; i-- > 0;) {
            F ui = F::mul(inv, pre[i]);                    // 1 / u_i
            inv = F::mul(inv, u[i]);
            t[i] = AT{F::mul(proj[i].x, F::mul(ui, proj[i].zzz)), F::mul(proj[i].y, F::mul(ui, proj[i].zz))};
        }
        return true;
    }
    PT mul(const u32 k[8]) const {
        int8_t d[65];
        recode16(k, d);
        PT acc = PT::inf();
        for (int j = 0; j < 65; j++) {
            if (d[j] > 0) madd(acc, t[j * 8 + d[j] - 1]);
            else if (d[j] < 0) madd(acc, neg_affine(t[j * 8 - d[j] - 1]));
        }
        return acc;
    }
};

// kp*P + kq*Q, one pass: 4 doublings + <= 2 additions per radix-16 digit
template <class F>
static XYZZ<F> joint_mul(const XYZZ<F> &P, const u32 kp[8], const XYZZ<F> &Q, const u32 kq[8]) {
    typedef XYZZ<F> PT;
    PT tp[8], tq[8];
    tp[0] = P;
    tq[0] = Q;
    for (int i = 1; i < 8; i++) {
        tp[i] = tp[i - 1]; add(tp[i], P);
        tq[i] = tq[i - 1]; add(tq[i], Q);
    }
    int8_t dp[65], dq[65];
    recode16(kp, dp);
    recode16(kq, dq);
    PT acc = PT::inf();
    for (int j = 64; j >= 0; j--) {
        if (!acc.is_inf()) for (int k = 0; k < 4; k++) acc = dbl(acc);
        if (dp[j] > 0) add(acc, tp[dp[j] - 1]); else if (dp[j] < 0) add(acc, neg(tp[-dp[j] - 1]));
        if (dq[j] > 0) add(acc, tq[dq[j] - 1]); else if (dq[j] < 0) add(acc, neg(tq[-dq[j] - 1]));
    }
    return acc;
}

// Tables of the last few keys seen (a server holds a handful of circuits; zk_assemble has no handle to keep
// them in).  Built on first use: 65*8 additions per group, about 1.5 ms, then 0.3 ms less per proof.
struct DeltaTables {
    uint8_t key[64 + 128];
    FixedBase<Fq64> d1;
    FixedBase<Fq2h> d2;
};
static std::shared_ptr<const DeltaTables> delta_tables(const uint8_t vk_delta1[64], const uint8_t vk_delta2[128]) {
    static std::mutex m;
    static std::list<std::shared_ptr<const DeltaTables>> lru;
    uint8_t key[192];
    memcpy(key, vk_delta1, 64);
    memcpy(key + 64, vk_delta2, 128);
    {
        std::lock_guard<std::mutex> lk(m);
        for (auto it = lru.begin(); it != lru.end(); ++it)
            if (!memcmp((*it)->key, key, 192)) {
                auto hit = *it;
                lru.erase(it);
                lru.push_front(hit);
                return hit;
            }
    }
    auto t = std::make_shared<DeltaTables>();          // built outside the lock: the threads of a first batched submission may each build it once
    memcpy(t->key, key, 192);
    t->d1.build(load<G1A>(vk_delta1));
    t->d2.build(load<G2A>(vk_delta2));
    std::lock_guard<std::mutex> lk(m);
    for (auto &e : lru)                                // ... but only ONE copy enters the cache: duplicates of a key evicted the tables of the server's other circuits
        if (!memcmp(e->key, key, 192)) return e;
    lru.push_front(t);
    if (lru.size() > 16) lru.pop_back();
    return t;
}

// A, B, C -> affine with ONE field inversion (the G2 one goes through the norm of zz*zzz)
static void three_to_affine(const G1P &a, const G2P &b, const G1P &c, G1A &A, G2A &B, G1A &C) {
    if (a.is_inf() || b.is_inf() || c.is_inf()) {       // never for a real proof; keep the encoding rules in one place
        A = to_affine(a);
        B = to_affine(b);
        C = to_affine(c);
        return;
    }
    const Fq64 ta = Fq64::mul(a.zz, a.zzz), tc = Fq64::mul(c.zz, c.zzz);
    const Fq2h tb = Fq2h::mul(b.zz, b.zzz);
    const Fq64 nb = Fq64::add(Fq64::sqr(tb.a), Fq64::sqr(tb.b));
    const Fq64 tac = Fq64::mul(ta, tc);
    const Fq64 inv = Fq64::inv(Fq64::mul(tac, nb));
    const Fq64 inb = Fq64::mul(inv, tac);                      // 1 / norm(tb)
    const Fq64 iac = Fq64::mul(inv, nb);                       // 1 / (ta tc)
    const Fq64 ia = Fq64::mul(iac, tc), ic = Fq64::mul(iac, ta);
    const Fq2h ib{Fq64::mul(tb.a, inb), Fq64::neg(Fq64::mul(tb.b, inb))};
    A = G1A{Fq64::mul(a.x, Fq64::mul(ia, a.zzz)), Fq64::mul(a.y, Fq64::mul(ia, a.zz))};
    C = G1A{Fq64::mul(c.x, Fq64::mul(ic, c.zzz)), Fq64::mul(c.y, Fq64::mul(ic, c.zz))};
    B = G2A{Fq2h::mul(b.x, Fq2h::mul(ib, b.zzz)), Fq2h::mul(b.y, Fq2h::mul(ib, b.zz))};
}

// what the tail needs of (r, s) and the key alone (HostTail::RsPart is this, opaque)
struct RsPoints {
    u32 r[8], s[8];
    G1P r_delta1, s_delta1, rs_delta1;
    G2P s_delta2;
};
static_assert(sizeof(RsPoints) <= sizeof(HostTail::RsPart) && alignof(RsPoints) <= 16, "RsPart too small");

static void rs_points(const uint8_t vk_delta1[64], const uint8_t vk_delta2[128], const uint8_t r32[32], const uint8_t s32[32], RsPoints &o) {
    u32 rs[8];
    memcpy(o.r, r32, 32);
    memcpy(o.s, s32, 32);
    // rs = toMontgomery(mul(r, s)) = r*s mod r_BN in standard form (:242-243)
    Fr64 fr, fs;
    memcpy(fr.v, r32, 32);
    memcpy(fs.v, s32, 32);
    Fr64 frs = Fr64::to_mont(Fr64::mul(fr, fs));
    memcpy(rs, frs.v, 32);

    const std::shared_ptr<const DeltaTables> tb = delta_tables(vk_delta1, vk_delta2);
    if (!tb->d1.t.empty()) {
        o.r_delta1 = tb->d1.mul(o.r);
        o.s_delta1 = tb->d1.mul(o.s);
        o.rs_delta1 = tb->d1.mul(rs);
    } else {
        const G1P delta1 = G1P::from_affine(load<G1A>(vk_delta1));
        o.r_delta1 = scalar_mul(delta1, o.r);
        o.s_delta1 = scalar_mul(delta1, o.s);
        o.rs_delta1 = scalar_mul(delta1, rs);
    }
    o.s_delta2 = !tb->d2.t.empty() ? tb->d2.mul(o.s) : scalar_mul(G2P::from_affine(load<G2A>(vk_delta2)), o.s);
}

static void assemble_core(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                          const G1P &pih, G1P pi_a, G1P pib1, G2P pi_b, G1P pi_c, const RsPoints &rsp,
                          uint8_t outA[64], uint8_t outB[128], uint8_t outC[64]) {
    const u32 *r = rsp.r, *s = rsp.s;
    const G1P &r_delta1 = rsp.r_delta1, &s_delta1 = rsp.s_delta1, &rs_delta1 = rsp.rs_delta1;
    const G2P &s_delta2 = rsp.s_delta2;

    madd(pi_a, load<G1A>(vk_alpha1));                       // groth16.cpp:222
    add(pi_a, r_delta1);                                    // :223-224
    madd(pi_b, load<G2A>(vk_beta2));                        // :226
    add(pi_b, s_delta2);                                    // :227-228
    madd(pib1, load<G1A>(vk_beta1));                        // :230
    add(pib1, s_delta1);                                    // :231-232
    add(pi_c, pih);                                         // :234
    add(pi_c, joint_mul(pi_a, s, pib1, r));                 // :236-240   s*pi_a + r*pi_b1
    add(pi_c, neg(rs_delta1));                              // :245-246
    G1A A, C;                                               // :249-251
    G2A B;
    three_to_affine(pi_a, pi_b, pi_c, A, B, C);
    memcpy(outA, &A, 64);
    memcpy(outB, &B, 128);
    memcpy(outC, &C, 64);
}

void HostTail::final_assembly(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                              const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                              const uint8_t pih_b[64], const uint8_t pi_a_b[64], const uint8_t pib1_b[64],
                              const uint8_t pi_b_b[128], const uint8_t pi_c_b[64],
                              const uint8_t r32[32], const uint8_t s32[32],
                              uint8_t outA[64], uint8_t outB[128], uint8_t outC[64]) {
    RsPoints rsp;
    rs_points(vk_delta1, vk_delta2, r32, s32, rsp);
    assemble_core(vk_alpha1, vk_beta1, vk_beta2, G1P::from_affine(load<G1A>(pih_b)),
                  G1P::from_affine(load<G1A>(pi_a_b)), G1P::from_affine(load<G1A>(pib1_b)), G2P::from_affine(load<G2A>(pi_b_b)),
                  G1P::from_affine(load<G1A>(pi_c_b)), rsp, outA, outB, outC);
}

static int draw31(uint8_t out[32]) {
    // src/groth16.cpp:213-217: zero, then 31 random bytes into the low bytes
    memset(out, 0, 32);
    size_t got = 0;
    while (got < 31) {
        ssize_t k = getrandom(out + got, 31 - got, 0);
        if (k < 0) return -1;
        got += (size_t)k;
    }
    return 0;
}

// The whole tail of an unsharded proof straight from the window sums the GPU wrote (XYZZ, device layout):
// no detour through the affine zk_msm_sums record (five inversions) that sharded provers exchange.
int HostTail::finish_from_windows(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                                  const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                                  const uint8_t *w1, const uint8_t *w2, uint32_t Ww, uint32_t cw, uint32_t rcw, uint32_t Wh, uint32_t ch, uint32_t rch,
                                  const uint8_t *r32, const uint8_t *s32, uint8_t outA[64], uint8_t outB[128], uint8_t outC[64],
                                  const RsPart *pre) {
    const size_t M1 = (size_t)Ww * rcw * sizeof(G1P);      // one MSM's records
    return finish_from_records(vk_alpha1, vk_beta1, vk_beta2, vk_delta1, vk_delta2, w1, w1 + M1, w1 + 2 * M1, w1 + 3 * M1, w2,
                               Ww, cw, rcw, Wh, ch, rch, r32, s32, outA, outB, outC, pre);
}

int HostTail::prepare_rs(const uint8_t vk_delta1[64], const uint8_t vk_delta2[128], const uint8_t *r32, const uint8_t *s32, RsPart *out) {
    uint8_t r[32], s[32];
    if (r32) memcpy(r, r32, 32); else if (draw31(r)) return 1;
    if (s32) memcpy(s, s32, 32); else if (draw31(s)) return 1;
    rs_points(vk_delta1, vk_delta2, r, s, *new (out->blob) RsPoints);
    return 0;
}

// the same with the five MSMs' records given one by one (a batched submission keeps one bucket set per proof)
int HostTail::finish_from_records(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                                  const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                                  const uint8_t *wa, const uint8_t *wb1, const uint8_t *wc, const uint8_t *wh, const uint8_t *wb2,
                                  uint32_t Ww, uint32_t cw, uint32_t rcw, uint32_t Wh, uint32_t ch, uint32_t rch,
                                  const uint8_t *r32, const uint8_t *s32, uint8_t outA[64], uint8_t outB[128], uint8_t outC[64],
                                  const RsPart *pre) {
    RsPoints mine;
    const RsPoints *rsp = pre ? reinterpret_cast<const RsPoints *>(pre->blob) : nullptr;
    if (!rsp) {
        uint8_t r[32], s[32];
        if (r32) memcpy(r, r32, 32); else if (draw31(r)) return 1;
        if (s32) memcpy(s, s32, 32); else if (draw31(s)) return 1;
        rs_points(vk_delta1, vk_delta2, r, s, mine);
        rsp = &mine;
    }
    G1P a, b1, c, h;
    G2P b2;
    if (Ww > 2 || Wh > 2) {            // plain tables: five Horner chains of W*c doublings, one host thread each (two sets — rows for every
                                       // second window — are c doublings per MSM: inline)
        std::thread t1([&] { a = horner<G1P>(wa, Ww, cw, rcw); });
        std::thread t2([&] { b1 = horner<G1P>(wb1, Ww, cw, rcw); });
        std::thread t3([&] { c = horner<G1P>(wc, Ww, cw, rcw); });
        std::thread t4([&] { h = horner<G1P>(wh, Wh, ch, rch); });
        b2 = horner<G2P>(wb2, Ww, cw, rcw);
        t1.join();
        t2.join();
        t3.join();
        t4.join();
    } else {
        a = horner<G1P>(wa, Ww, cw, rcw);
        b1 = horner<G1P>(wb1, Ww, cw, rcw);
        c = horner<G1P>(wc, Ww, cw, rcw);
        h = horner<G1P>(wh, Wh, ch, rch);
        b2 = horner<G2P>(wb2, Ww, cw, rcw);
    }
    assemble_core(vk_alpha1, vk_beta1, vk_beta2, h, a, b1, b2, c, *rsp, outA, outB, outC);
    return 0;
}

std::string HostTail::to_dec(const uint8_t le32[32]) {
    uint32_t w[8];
    memcpy(w, le32, 32);
    // repeated division by 10^9
    char buf[96];
    int pos = 96;
    buf[--pos] = 0;
    bool nz = false;
    for (int i = 0; i < 8; i++) nz |= (w[i] != 0);
    if (!nz) return "0";
    std::vector<uint32_t> chunks;
    while (true) {
        bool any = false;
        uint64_t rem = 0;
        for (int i = 7; i >= 0; i--) {
            uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 1000000000u);
            rem = cur % 1000000000u;
            any |= (w[i] != 0);
        }
        chunks.push_back((uint32_t)rem);
        if (!any) break;
    }
    std::string out;
    char tmp[16];
    snprintf(tmp, sizeof tmp, "%u", chunks.back());
    out += tmp;
    for (int i = (int)chunks.size() - 2; i >= 0; i--) {
        snprintf(tmp, sizeof tmp, "%09u", chunks[i]);
        out += tmp;
    }
    return out;
}

std::string HostTail::fq_mont_to_dec(const uint8_t le32[32]) {
    Fq64 x;
    memcpy(x.v, le32, 32);
    Fq64 s = Fq64::from_mont(x);
    return to_dec((const uint8_t *)s.v);
}

}   // namespace zk

// ------------------------------------------------------------------ C ABI (host-only part)
extern "C" {

const char *zk_last_error(void) { return zk::get_error(); }

static size_t emit(const std::string &s, char *buf, size_t cap) {
    if (buf && cap) {
        size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

int zk_g1_mul(uint8_t out[64], const uint8_t p[64], const uint8_t k[32]) {
    zk::u32 kk[8];
    memcpy(kk, k, 32);
    zk::G1A a = zk::load<zk::G1A>(p);
    zk::G1A r = zk::to_affine(zk::scalar_mul(zk::G1P::from_affine(a), kk));
    memcpy(out, &r, 64);
    return 0;
}
int zk_g2_mul(uint8_t out[128], const uint8_t p[128], const uint8_t k[32]) {
    zk::u32 kk[8];
    memcpy(kk, k, 32);
    zk::G2A a = zk::load<zk::G2A>(p);
    zk::G2A r = zk::to_affine(zk::scalar_mul(zk::G2P::from_affine(a), kk));
    memcpy(out, &r, 128);
    return 0;
}


int zk_assemble(const void *vk_alpha1, const void *vk_beta1, const void *vk_beta2, const void *vk_delta1,
                const void *vk_delta2, const zk_msm_sums *parts, uint32_t nparts, const uint8_t *r32,
                const uint8_t *s32, zk_proof *out) {
    if (!vk_alpha1 || !vk_beta1 || !vk_beta2 || !vk_delta1 || !vk_delta2 || !parts || !out || !nparts) {
        zk::set_error("zk_assemble: null argument or no partial sums");
        return 1;
    }
    zk_msm_sums t = parts[0];
    for (uint32_t i = 1; i < nparts; i++) {
        zk::HostTail::add_affine_g1(t.pih, parts[i].pih);
        zk::HostTail::add_affine_g1(t.pi_a, parts[i].pi_a);
        zk::HostTail::add_affine_g1(t.pib1, parts[i].pib1);
        zk::HostTail::add_affine_g2(t.pi_b, parts[i].pi_b);
        zk::HostTail::add_affine_g1(t.pi_c, parts[i].pi_c);
    }
    uint8_t r[32], s[32];
    if (r32) memcpy(r, r32, 32); else if (zk::draw31(r)) { zk::set_error("getrandom failed"); return 1; }
    if (s32) memcpy(s, s32, 32); else if (zk::draw31(s)) { zk::set_error("getrandom failed"); return 1; }
    zk::HostTail::final_assembly((const uint8_t *)vk_alpha1, (const uint8_t *)vk_beta1, (const uint8_t *)vk_beta2,
                                 (const uint8_t *)vk_delta1, (const uint8_t *)vk_delta2, t.pih, t.pi_a, t.pib1, t.pi_b,
                                 t.pi_c, r, s, out->A, out->B, out->C);
    return 0;
}

size_t zk_proof_to_json(const zk_proof *p, char *buf, size_t cap) {
    // Key order / spacing of nlohmann's compact dump of Proof::toJson
    auto d = [](const uint8_t *b) { return zk::HostTail::fq_mont_to_dec(b); };
    std::string s = "{\"pi_a\":[\"" + d(p->A) + "\",\"" + d(p->A + 32) + "\",\"1\"],\"pi_b\":[[\"" + d(p->B) + "\",\"" +
                    d(p->B + 32) + "\"],[\"" + d(p->B + 64) + "\",\"" + d(p->B + 96) +
                    "\"],[\"1\",\"0\"]],\"pi_c\":[\"" + d(p->C) + "\",\"" + d(p->C + 32) +
                    "\",\"1\"],\"protocol\":\"groth16\"}";
    return emit(s, buf, cap);
}

size_t zk_public_to_json(const uint8_t *wtns, uint32_t nPublic, char *buf, size_t cap) {
    if (nPublic == 0) return emit("null", buf, cap);   // main_prover.cpp:85-92 (quirk Q7)
    std::string s = "[";
    for (uint32_t i = 1; i <= nPublic; i++) {
        if (i > 1) s += ",";
        s += "\"" + zk::HostTail::to_dec(wtns + (size_t)i * 32) + "\"";
    }
    s += "]";
    return emit(s, buf, cap);
}

}   // extern "C"
