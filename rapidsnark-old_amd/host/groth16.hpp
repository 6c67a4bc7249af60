// Groth16::Prover / makeProver / Proof — the reference's C++ seam (src/groth16.hpp:13-121)
// re-created over the C-ABI of libzkhip.so.  The Engine template parameter is gone: there is
// one engine, BN254 on an MI355X.  Same call shapes:
//     auto prover = Groth16::makeProver(nVars, nPublic, domainSize, nCoefs, vk..., sections...);
//     auto proof  = prover->prove(wtnsData);        // src/main_prover.cpp:57-75
//     out << proof->toJson();                       // compact JSON, bytes of SURVEY §A.3
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../include/zkhip.h"

namespace Groth16 {

class Proof {
public:
    zk_proof raw;   // A|B|C affine Montgomery — the bytes of the reference's Proof<Engine>
    std::string toJson() const {
        size_t n = zk_proof_to_json(&raw, nullptr, 0);
        std::string s(n + 1, '\0');
        zk_proof_to_json(&raw, &s[0], n + 1);
        s.resize(n);
        return s;
    }
};

class Prover {
    zk_prover *h_ = nullptr;

public:
    explicit Prover(zk_prover *h) : h_(h) {}
    ~Prover() { zk_prover_destroy(h_); }
    Prover(const Prover &) = delete;
    Prover &operator=(const Prover &) = delete;

    // wtns: nVars x 32 B standard form (section 2 of the .wtns).  r/s: optional fixed scalars
    // (32 B LE) replacing the reference's randombytes_buf (src/groth16.cpp:216-217).
    std::unique_ptr<Proof> prove(const void *wtns, const uint8_t *r32 = nullptr, const uint8_t *s32 = nullptr) {
        std::unique_ptr<Proof> p(new Proof());
        if (zk_prove(h_, static_cast<const uint8_t *>(wtns), r32, s32, &p->raw) != 0) throw std::runtime_error(zk_last_error());
        return p;
    }
};

inline std::unique_ptr<Prover> makeProver(uint32_t nVars, uint32_t nPublic, uint32_t domainSize, uint64_t nCoefs,
                                          void *vk_alpha1, void *vk_beta1, void *vk_beta2, void *vk_delta1, void *vk_delta2,
                                          void *coefs, void *pointsA, void *pointsB1, void *pointsB2, void *pointsC,
                                          void *pointsH, const uint64_t sectionBytes[6] = nullptr, bool precompDefault = false) {
    zk_zkey_view v{};
    v.nVars = nVars;
    v.nPublic = nPublic;
    v.domainSize = domainSize;
    v.nCoefs = nCoefs;
    v.vk_alpha1 = vk_alpha1;
    v.vk_beta1 = vk_beta1;
    v.vk_beta2 = vk_beta2;
    v.vk_delta1 = vk_delta1;
    v.vk_delta2 = vk_delta2;
    v.coefs = coefs;
    v.pointsA = pointsA;
    v.pointsB1 = pointsB1;
    v.pointsB2 = pointsB2;
    v.pointsC = pointsC;
    v.pointsH = pointsH;
    if (sectionBytes) {
        v.coefs_bytes = sectionBytes[0];
        v.pointsA_bytes = sectionBytes[1];
        v.pointsB1_bytes = sectionBytes[2];
        v.pointsB2_bytes = sectionBytes[3];
        v.pointsC_bytes = sectionBytes[4];
        v.pointsH_bytes = sectionBytes[5];
    }
    zk_opts o{};
    o.device = -1;
    // ZKHIP_PRECOMP=1/0: window-precomputed tables (W x table memory, longer create, ~10 % faster
    // proofs).  Default: off for the one-shot CLI, on for the server where create is amortised.
    const char *pc = getenv("ZKHIP_PRECOMP");
    if (pc ? (pc[0] == '1') : precompDefault) o.flags |= ZK_FLAG_PRECOMP;
    if (const char *dev = getenv("ZKHIP_DEVICE")) o.device = atoi(dev);
    zk_prover *h = nullptr;
    if (zk_prover_create(&h, &v, &o) != 0) throw std::runtime_error(zk_last_error());
    return std::unique_ptr<Prover>(new Prover(h));
}

}   // namespace Groth16
