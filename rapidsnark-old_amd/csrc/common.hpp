// Shared host-side plumbing of libzkhip: error reporting across the C ABI and the
// host-tail entry points (compiled by g++ in host_tail.cpp, called from prover_pipeline.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <stdlib.h>

namespace zk {

// Measurement probes (chunk sizes, gather confinement, workgroup rounds ...) exist only in a build made with
// -DZK_PROBES (make probes -> libzkhip_probes.so, loaded through ZKHIP_LIB for same-box A/Bs).  The shipped
// library reads the variables INTEGRATION.md section 5 lists and nothing else: probe_env() is a constant there.
static inline const char *probe_env(const char *name) {
#ifdef ZK_PROBES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// thread-local message behind zk_last_error()
void set_error(const std::string &msg);
const char *get_error();

struct HostTail {
    // Window-sum Horner: out = sum_w 2^(c*w) * windows[w]; windows are XYZZ in device layout
    // (G1: 4 x 32 B, G2: 4 x 64 B per point).  Result affine Montgomery (zero = infinity).
    // rc records per window (kernels.hpp, msm_wsum_rc): 1 = the window's sum, c = the bit sums T, S_0 .. S_{c-2}
    static void combine_windows_g1(const uint8_t *windows_xyzz, uint32_t W, uint32_t c, uint32_t rc, uint8_t out_affine[64]);
    static void combine_windows_g2(const uint8_t *windows_xyzz, uint32_t W, uint32_t c, uint32_t rc, uint8_t out_affine[128]);
    // out += in (affine Montgomery, zero = infinity): adds shard partial sums
    static void add_affine_g1(uint8_t acc[64], const uint8_t in[64]);
    static void add_affine_g2(uint8_t acc[128], const uint8_t in[128]);
    // src/groth16.cpp:219-251 — final assembly from the five MSM results.
    static void final_assembly(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                               const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                               const uint8_t pih[64], const uint8_t pi_a[64], const uint8_t pib1[64],
                               const uint8_t pi_b[128], const uint8_t pi_c[64],
                               const uint8_t r32[32], const uint8_t s32[32],
                               uint8_t outA[64], uint8_t outB[128], uint8_t outC[64]);
    // The part of the tail that needs (r, s) and the key but NOT the MSM sums — r*delta1, s*delta1, rs*delta1, s*delta2, about
    // 0.13 of the tail's 0.3 ms — so that a collecting thread can do it BEFORE it waits for the GPU.  Opaque (the point types are
    // host_tail.cpp's).  prepare_rs draws r32 / s32 when they are NULL; nonzero = the random source failed.
    struct RsPart {
        alignas(16) uint8_t blob[800];
    };
    static int prepare_rs(const uint8_t vk_delta1[64], const uint8_t vk_delta2[128], const uint8_t *r32, const uint8_t *s32, RsPart *out);
    // the same from the GPU's window sums (XYZZ; G1: A, B1, C [Ww each] then H [Wh]; G2: B2 [Ww]); r32 / s32 NULL = drawn
    // like the reference's randombytes_buf(31 bytes).  Returns nonzero when the random source fails.
    static int finish_from_windows(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                                   const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                                   const uint8_t *w1, const uint8_t *w2, uint32_t Ww, uint32_t cw, uint32_t rcw, uint32_t Wh, uint32_t ch, uint32_t rch,
                                   const uint8_t *r32, const uint8_t *s32, uint8_t outA[64], uint8_t outB[128], uint8_t outC[64],
                                   const RsPart *pre = nullptr);        // pre given: (r32, s32) are ignored — they are in it
    static int finish_from_records(const uint8_t vk_alpha1[64], const uint8_t vk_beta1[64], const uint8_t vk_beta2[128],
                                   const uint8_t vk_delta1[64], const uint8_t vk_delta2[128],
                                   const uint8_t *wa, const uint8_t *wb1, const uint8_t *wc, const uint8_t *wh, const uint8_t *wb2,
                                   uint32_t Ww, uint32_t cw, uint32_t rcw, uint32_t Wh, uint32_t ch, uint32_t rch,
                                   const uint8_t *r32, const uint8_t *s32, uint8_t outA[64], uint8_t outB[128], uint8_t outC[64],
                                   const RsPart *pre = nullptr);
    // canonical base-10 of a 32-byte LE integer
    static std::string to_dec(const uint8_t le32[32]);
    // de-Montgomery an Fq element and print base-10 (E.f1.toString, src/groth16.cpp:274)
    static std::string fq_mont_to_dec(const uint8_t le32[32]);
};

}   // namespace zk

// Wave priority (s_setprio) by what a kernel is — the level-1 bucket accumulations, which fill the chip for milliseconds, stay at 0:
//   3  the follow-up kernels (merges of the cut runs, bucket reductions): chains of dependent additions in a handful of waves;
//      beside a level-1 launch each of their waves shares its SIMD's issue port with three level-1 waves
//   1  the kernels of the two sort chains and of the transform chain (SpMV, passes, abc -> h): what the NEXT level-1 launch of
//      their proof waits for
// Queue / stream priorities only order the DISPATCH of workgroups (measured earlier: nothing for unsharded provers); this is the
// SIMD's issue arbiter.  Same box, three alternations (profiles/r04bg_*, r04bh_*): pipelined period 2^14 0.612 -> 0.567 ms,
// 2^16 1.005 -> 0.938, 2^18 2.81 -> 2.62, 2^20 9.18 -> 8.95, 2^22 32.46 -> 32.30 (witness resident; with host witnesses the same
// ratios); synchronous proofs and shards equal.  Follow-ups alone: -5 ... -7 % up to 2^18, nothing above; chains at 2 instead
// of 1: equal.  -DZK_TAIL_WAVE_PRIO=0 -DZK_CHAIN_WAVE_PRIO=0 builds the library without.
#ifndef ZK_TAIL_WAVE_PRIO
#define ZK_TAIL_WAVE_PRIO 3
#endif
#if ZK_TAIL_WAVE_PRIO
#define ZK_TAIL_PRIO() __builtin_amdgcn_s_setprio(ZK_TAIL_WAVE_PRIO)
#else
#define ZK_TAIL_PRIO() ((void)0)
#endif
#ifndef ZK_CHAIN_WAVE_PRIO
#define ZK_CHAIN_WAVE_PRIO 1
#endif
#if ZK_CHAIN_WAVE_PRIO
#define ZK_CHAIN_PRIO() __builtin_amdgcn_s_setprio(ZK_CHAIN_WAVE_PRIO)
#else
#define ZK_CHAIN_PRIO() ((void)0)
#endif
