// Host tail of a proof: window Horner combine, shard-sum, final assembly, JSON.
// Compiled by g++ (no HIP).  Mirrors src/groth16.cpp:209-301 and src/main_prover.cpp:77-93.
#include "common.hpp"
#include "field64.hpp"
#include "curve.hpp"
#include "../../include/zkhip.h"

#include <string.h>
#include <stdio.h>
#include <sys/random.h>
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

template <class PT, class AT>
static void combine_windows(const uint8_t *w, uint32_t W, uint32_t c, uint8_t *out) {
    PT acc = PT::inf();
    for (int i = (int)W - 1; i >= 0; i--) {
        for (uint32_t k = 0; k < c; k++) acc = dbl(acc);
        PT s;
        memcpy(&s, w + (size_t)i * sizeof(PT), sizeof(PT));
        add(acc, s);
    }
    AT a = to_affine(acc);
    memcpy(out, &a, sizeof(AT));
}
void HostTail::combine_windows_g1(const uint8_t *w, uint32_t W, uint32_t c, uint8_t out[64]) {
    combine_windows<G1P, G1A>(w, W, c, out);
}
void HostTail::combine_windows_g2(const uint8_t *w, uint32_t W, uint32_t c, uint8_t out[128]) {
    combine_windows<G2P, G2A>(w, W, c, out);
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

void HostTail::final_assembly(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                              const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                              const uint8_t pih_b[64], const uint8_t pi_a_b[64], const uint8_t pib1_b[64],
                              const uint8_t pi_b_b[128], const uint8_t pi_c_b[64],
                              const uint8_t r32[32], const uint8_t s32[32],
                              uint8_t outA[64], uint8_t outB[128], uint8_t outC[64]) {
    u32 r[8], s[8], rs[8];
    memcpy(r, r32, 32);
    memcpy(s, s32, 32);
    G1P pi_a = G1P::from_affine(load<G1A>(pi_a_b));
    G1P pib1 = G1P::from_affine(load<G1A>(pib1_b));
    G2P pi_b = G2P::from_affine(load<G2A>(pi_b_b));
    G1P pi_c = G1P::from_affine(load<G1A>(pi_c_b));
    G1P pih = G1P::from_affine(load<G1A>(pih_b));
    G1P delta1 = G1P::from_affine(load<G1A>(vk_delta1));
    G2P delta2 = G2P::from_affine(load<G2A>(vk_delta2));

    // rs = toMontgomery(mul(r, s)) = r*s mod r_BN in standard form (:242-243)
    Fr64 fr, fs;
    memcpy(fr.v, r32, 32);
    memcpy(fs.v, s32, 32);
    Fr64 frs = Fr64::to_mont(Fr64::mul(fr, fs));
    memcpy(rs, frs.v, 32);

    // The six 256-bit scalar multiplications dominate this tail (~0.2 ms each in G1, ~0.55 ms in
    // G2 on one core).  Four are independent of the MSM results, two more depend only on A / B1:
    // two short waves of host threads instead of the reference's serial chain (:222-246).
    G1P r_delta1, s_delta1, rs_delta1, s_A, r_B1;
    G2P s_delta2;
    {
        std::thread t1([&] { r_delta1 = scalar_mul(delta1, r); });
        std::thread t2([&] { s_delta1 = scalar_mul(delta1, s); });
        std::thread t3([&] { rs_delta1 = scalar_mul(delta1, rs); });
        s_delta2 = scalar_mul(delta2, s);
        t1.join();
        t2.join();
        t3.join();
    }
    madd(pi_a, load<G1A>(vk_alpha1));                       // groth16.cpp:222
    add(pi_a, r_delta1);                                    // :223-224
    madd(pi_b, load<G2A>(vk_beta2));                        // :226
    add(pi_b, s_delta2);                                    // :227-228
    madd(pib1, load<G1A>(vk_beta1));                        // :230
    add(pib1, s_delta1);                                    // :231-232
    {
        std::thread t1([&] { s_A = scalar_mul(pi_a, s); });
        r_B1 = scalar_mul(pib1, r);
        t1.join();
    }
    add(pi_c, pih);                                         // :234
    add(pi_c, s_A);                                         // :236-237
    add(pi_c, r_B1);                                        // :239-240
    add(pi_c, neg(rs_delta1));                              // :245-246
    G1A A = to_affine(pi_a);                                // :249-251
    G2A B = to_affine(pi_b);
    G1A C = to_affine(pi_c);
    memcpy(outA, &A, 64);
    memcpy(outB, &B, 128);
    memcpy(outC, &C, 64);
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
    if (r32) memcpy(r, r32, 32); else if (draw31(r)) { zk::set_error("getrandom failed"); return 1; }
    if (s32) memcpy(s, s32, 32); else if (draw31(s)) { zk::set_error("getrandom failed"); return 1; }
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
