// Groth16::Prover / makeProver / Proof — the reference's C++ seam (src/groth16.hpp:13-121)
// re-created over the C-ABI of libzkhip.so.  The Engine template parameter is gone: there is
// one engine, BN254 on an MI355X.  Same call shapes:
//     auto prover = Groth16::makeProver(nVars, nPublic, domainSize, nCoefs, vk..., sections...);
//     auto proof  = prover->prove(wtnsData);        // src/main_prover.cpp:57-75
//     out << proof->toJson();                       // compact JSON, bytes of SURVEY §A.3
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

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
    zk_multi_prover *m_ = nullptr;      // ZKHIP_DEVICES=0,1,...: one proof split over several GPUs
    uint32_t batch_ = 1;                // witnesses one submission may carry (zk_opts.batch)
    uint32_t reserved_ = 0;             // proofs in flight whose workspace was allocated at creation (zk_prover_reserve); 0 = none asked for

public:
    explicit Prover(zk_prover *h, uint32_t batch = 1) : h_(h), batch_(batch > 1 ? batch : 1) {}
    explicit Prover(zk_multi_prover *m) : m_(m) {}
    ~Prover() {
        if (h_) zk_prover_destroy(h_);
        if (m_) zk_multi_prover_destroy(m_);
    }
    Prover(const Prover &) = delete;
    Prover &operator=(const Prover &) = delete;

    // wtns: nVars x 32 B standard form (section 2 of the .wtns).  r/s: optional fixed scalars
    // (32 B LE) replacing the reference's randombytes_buf (src/groth16.cpp:216-217).
    std::unique_ptr<Proof> prove(const void *wtns, const uint8_t *r32 = nullptr, const uint8_t *s32 = nullptr) {
        std::unique_ptr<Proof> p(new Proof());
        const auto *w = static_cast<const uint8_t *>(wtns);
        const int rc = m_ ? zk_multi_prove(m_, w, r32, s32, &p->raw) : zk_prove(h_, w, r32, s32, &p->raw);
        if (rc != 0) throw std::runtime_error(zk_last_error());
        return p;
    }
    // Throughput mode (not in the reference, which proves one at a time): up to ZK_MAX_IN_FLIGHT proofs
    // enqueued; `wtns` must stay valid until the matching collect().
    void submit(const void *wtns, const uint8_t *r32 = nullptr, const uint8_t *s32 = nullptr) {
        const auto *w = static_cast<const uint8_t *>(wtns);
        if ((m_ ? zk_multi_prove_submit(m_, w, r32, s32) : zk_prove_submit(h_, w, r32, s32)) != 0) throw std::runtime_error(zk_last_error());
    }
    std::unique_ptr<Proof> collect() {
        std::unique_ptr<Proof> p(new Proof());
        if ((m_ ? zk_multi_prove_collect(m_, &p->raw) : zk_prove_collect(h_, &p->raw)) != 0) throw std::runtime_error(zk_last_error());
        return p;
    }
    // Small circuits: up to batch() witnesses of the circuit in ONE submission (zk_prove_batch_*): one set of kernel
    // launches for all of them.  r32 / s32 (optional) are used for every proof of the submission.
    uint32_t batch() const { return batch_; }
    // makeProver(..., reserveInFlight): what could be reserved — the depth asked for, or 1 where that did not fit
    uint32_t reservedInFlight() const { return reserved_; }
    void setReserved(uint32_t n) { reserved_ = n; }
    void submitBatch(const std::vector<const void *> &wtns, const uint8_t *r32 = nullptr, const uint8_t *s32 = nullptr) {
        if (wtns.empty() || wtns.size() > batch_) throw std::invalid_argument("submitBatch: between 1 and batch() witnesses");
        if (m_) throw std::invalid_argument("submitBatch on a multi-GPU prover");
        std::vector<const uint8_t *> ptrs;
        std::vector<uint8_t> rr, ss;
        for (const void *w : wtns) {
            ptrs.push_back(static_cast<const uint8_t *>(w));
            if (r32) rr.insert(rr.end(), r32, r32 + 32);
            if (s32) ss.insert(ss.end(), s32, s32 + 32);
        }
        if (zk_prove_batch_submit(h_, ptrs.data(), (uint32_t)ptrs.size(), r32 ? rr.data() : nullptr, s32 ? ss.data() : nullptr) != 0)
            throw std::runtime_error(zk_last_error());
    }
    std::vector<std::unique_ptr<Proof>> collectBatch(uint32_t count) {
        std::vector<zk_proof> raw(count);
        if (zk_prove_batch_collect(h_, raw.data(), count) != 0) throw std::runtime_error(zk_last_error());
        std::vector<std::unique_ptr<Proof>> out;
        for (uint32_t k = 0; k < count; k++) {
            out.emplace_back(new Proof());
            out.back()->raw = raw[k];
        }
        return out;
    }
};

// makeProver(..., reserveInFlight = kLibraryDepth): reserve the pipeline depth zk_prover_info recommends for the prover just created
// (host witnesses) instead of a number the caller derived from the circuit's size
static const uint32_t kLibraryDepth = 0xFFFFFFFFu;

// "0,1,2,3" -> {0,1,2,3}
inline std::vector<int32_t> parseDeviceList(const char *s) {
    std::vector<int32_t> v;
    while (s && *s) {
        char *end = nullptr;
        long d = strtol(s, &end, 10);
        if (end == s) throw std::invalid_argument("ZKHIP_DEVICES must be a comma-separated list of device ordinals");
        v.push_back((int32_t)d);
        s = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') throw std::invalid_argument("ZKHIP_DEVICES must be a comma-separated list of device ordinals");
    }
    return v;
}

inline std::unique_ptr<Prover> makeProver(uint32_t nVars, uint32_t nPublic, uint32_t domainSize, uint64_t nCoefs,
                                          void *vk_alpha1, void *vk_beta1, void *vk_beta2, void *vk_delta1, void *vk_delta2,
                                          void *coefs, void *pointsA, void *pointsB1, void *pointsB2, void *pointsC,
                                          void *pointsH, const uint64_t sectionBytes[6] = nullptr, bool precompDefault = false,
                                          int device = -1, uint32_t batch = 0, uint32_t reserveInFlight = 0) {
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
    // ZKHIP_PRECOMP=1/2/0: window-precomputed tables (W x table memory, longer create, ~13 % faster proofs); 2: rows for every
    // second window only (ceil(W/2) x the memory, same additions per point, two bucket reductions per MSM: ZK_FLAG_PRECOMP_HALF).
    // Default: off for the one-shot CLI, on for the server where create is amortised.
    const char *pc = getenv("ZKHIP_PRECOMP");
    if (pc ? (pc[0] == '1' || pc[0] == '2') : precompDefault) o.flags |= ZK_FLAG_PRECOMP;
    if (pc && pc[0] == '2') o.flags |= ZK_FLAG_PRECOMP_HALF;
    // ZKHIP_SPARSE_WITNESS=1 (with precomputed tables): 16-bit window for the four witness MSMs — for deployments whose witnesses
    // are circuit witnesses (mostly 0, 1 and small values), ZK_FLAG_SPARSE_WITNESS in include/zkhip.h
    if (const char *sw = getenv("ZKHIP_SPARSE_WITNESS"))
        if (sw[0] == '1') o.flags |= ZK_FLAG_SPARSE_WITNESS;
    if (device >= 0) o.device = device;
    else if (const char *dev = getenv("ZKHIP_DEVICE")) o.device = atoi(dev);
    // ZKHIP_DEVICES=0,1,...,7: ONE proof over several GPUs of the node — every MSM table sharded by point
    // range, the A.w/B.w rows and the transforms partitioned the same way (zk_multi_prover).  Only when no
    // explicit device was asked for (the server's per-GPU workers pass one).
    if (device < 0) {
        const std::vector<int32_t> devs = parseDeviceList(getenv("ZKHIP_DEVICES"));
        if (devs.size() > 1) {
            zk_multi_prover *m = nullptr;
            if (zk_multi_prover_create(&m, &v, devs.data(), (uint32_t)devs.size(), &o) != 0) throw std::runtime_error(zk_last_error());
            return std::unique_ptr<Prover>(new Prover(m));
        }
        if (devs.size() == 1) o.device = devs[0];
    }
    if (batch > 1 && (o.flags & ZK_FLAG_PRECOMP)) o.batch = batch > ZK_MAX_BATCH ? ZK_MAX_BATCH : batch;
    zk_prover *h = nullptr;
    // reserveInFlight (proverServer): the workspace of every proof slot and lane a pipeline of that depth walks is allocated
    // NOW (zk_prover_reserve) — slots and lanes otherwise appear when a depth is first reached, and a GPU whose tables
    // nearly fill its memory would create successfully and fail proofs later
    uint32_t reserved = 0;
    auto oom = [] { return strstr(zk_last_error(), "out of memory") != nullptr; };
    // create, then reserve the pipeline: the depth asked for, else (out of memory) two in flight on the SAME prover before
    // anything is given up — a shallower pipeline over window-precomputed tables (13 instead of 16 additions per point) is
    // faster than a deep one over the tables as in the zkey (2^22: 36.8 ms at two in flight against 38.4 at six)
    auto create = [&](uint32_t reserve) {
        int rc = zk_prover_create(&h, &v, &o);
        if (rc != 0 || !reserve) return rc;
        if (reserve == kLibraryDepth) {          // the depth that saturates THIS prover, as the library planned it (zk_prover_info)
            zk_prover_plan plan;
            memset(&plan, 0, sizeof plan);
            plan.size = sizeof plan;
            reserve = zk_prover_info(h, &plan) == 0 && plan.depth_host_witness ? plan.depth_host_witness : 3;
        }
        rc = zk_prover_reserve(h, reserve, 1);
        if (rc == 0) {
            reserved = reserve;
            return 0;
        }
        if (reserve > 2 && oom()) {
            std::cerr << "the workspace of " << reserve << " proofs in flight does not fit the GPU's free memory: two in flight\n";
            rc = zk_prover_reserve(h, 2, 1);
            if (rc == 0) {
                reserved = 2;
                return 0;
            }
        }
        zk_prover_destroy(h);
        h = nullptr;
        return rc;
    };
    int rc = create(reserveInFlight);
    // window-precomputed tables are 13 x the table memory (2^25 constraints: 167 GB + workspace; 2^26: > 288 GB): where they
    // were only the DEFAULT (proverServer, or a server that already holds other keys) and they, or the workspace of even two
    // proofs beside them, do not fit: first the rows of every second window (7 x, the same 13 additions per point) ...
    if (rc != 0 && (o.flags & ZK_FLAG_PRECOMP) && !(o.flags & ZK_FLAG_PRECOMP_HALF) && !pc && oom()) {
        std::cerr << "window-precomputed tables do not fit the GPU's free memory: rows for every second window\n";
        o.flags |= ZK_FLAG_PRECOMP_HALF;
        o.batch = 0;                               // (batched submissions need a row per window)
        rc = create(reserveInFlight);
        if (rc != 0) o.flags &= ~(uint32_t)ZK_FLAG_PRECOMP_HALF;
    }
    // ... then the tables as they are in the zkey instead of no prover at all
    if (rc != 0 && (o.flags & ZK_FLAG_PRECOMP) && !pc && oom()) {
        std::cerr << "window-precomputed tables do not fit the GPU's free memory: using the tables as in the zkey\n";
        o.flags &= ~(uint32_t)ZK_FLAG_PRECOMP;
        o.batch = 0;
        rc = create(reserveInFlight);
    }
    // ... and where even then the workspace of a pipeline does not fit: one proof at a time (the caller reads the depth
    // it may use from reservedInFlight())
    if (rc != 0 && reserveInFlight > 1 && oom()) {      // (kLibraryDepth included)
        std::cerr << "the workspace of two proofs in flight does not fit the GPU's free memory: one proof at a time\n";
        rc = create(1);
    }
    if (rc != 0) throw std::runtime_error(zk_last_error());
    std::unique_ptr<Prover> pr(new Prover(h, o.batch));
    pr->setReserved(reserved);
    return pr;
}

}   // namespace Groth16
