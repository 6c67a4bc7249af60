// BN254 prime-field arithmetic for gfx950 (and the host, for the O(1) final assembly).
//
// Replaces the reference's ffiasm RawFr/RawFq x86-64 ADX assembly (absent submodule,
// reference .gitmodules:7-9; call sites src/groth16.cpp:71-82,91-95,109,160-162).
// Representation: 8 x 32-bit limbs, little-endian, Montgomery form with R = 2^256 —
// byte-identical to the reference's 4 x 64-bit FrElement/FqElement, so zkey/wtns
// sections are used in place.
//
// This 8x32-bit saturated form is the CONTAINER type of every HBM-resident element and the
// arithmetic of the one-off / tool kernels (twiddle build, synthetic chains, zk_f*_mul_vec).
// The hot kernels compute in 9x29-bit signed limbs instead (field29.hpp): on gfx950 a
// v_addc_co_u32 costs as much as a v_mad_u64_u32, and this CIOS product needs ~250 of them
// next to its 128 MADs (measured 87.8 G products/s vs 174 G for the 29-bit form).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

typedef uint32_t u32;
typedef uint64_t u64;

struct FrParams {
    // r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (main_prover.cpp:34)
    static constexpr u32 P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                  0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};   // R mod r
    static constexpr u32 R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                  0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};   // R^2 mod r
    static constexpr u32 N0INV = 0xefffffffu;   // -r^-1 mod 2^32
};

struct FqParams {
    // q = 21888242871839275222246405745257275088696311157297823662689037894645226208583 (tasksfile.js:10)
    static constexpr u32 P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                  0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};   // R mod q
    static constexpr u32 R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                  0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};   // R^2 mod q
    static constexpr u32 N0INV = 0xe4866389u;   // -q^-1 mod 2^32
};

// a + b + carry-in -> (sum, carry-out).  __builtin_addc/__builtin_subc lower to a clean
// v_add_co_u32 / v_addc_co_u32 (v_sub_co / v_subb_co) chain on gfx950; the (u64) idiom does not.
#if defined(__clang__)
ZK_HD u32 addc(u32 a, u32 b, u32 &carry) {
    u32 co;
    u32 s = __builtin_addc(a, b, carry, &co);
    carry = co;
    return s;
}
ZK_HD u32 subb(u32 a, u32 b, u32 &borrow) {
    u32 bo;
    u32 d = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return d;
}
#else   // g++ host build (host_tail.cpp only touches the 32-bit Fp for layout typedefs)
ZK_HD u32 addc(u32 a, u32 b, u32 &carry) {
    u64 s = (u64)a + b + carry;
    carry = (u32)(s >> 32);
    return (u32)s;
}
ZK_HD u32 subb(u32 a, u32 b, u32 &borrow) {
    u64 d = (u64)a - b - borrow;
    borrow = (u32)(d >> 32) & 1u;
    return (u32)d;
}
#endif

template <class PR>
struct Fp {
    u32 v[8];

    ZK_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fp one() {   // Montgomery form of 1
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = PR::R1[i];
        return r;
    }
    ZK_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = PR::R2[i];
        return r;
    }
    ZK_HD bool is_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i];
        return o == 0;
    }
    ZK_HD bool is_zero_raw() const { return is_zero(); }   // canonical representation: same test
    ZK_HD bool operator==(const Fp &b) const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Fp &b) const { return !(*this == b); }

    // r = a - p if a >= p else a   (a < 2p)
    ZK_HD static Fp reduce_once(const Fp &a) {
        Fp d;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb(a.v[i], PR::P[i], bw);
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = bw ? a.v[i] : d.v[i];
        return r;
    }

    ZK_HD static Fp add(const Fp &a, const Fp &b) {
        Fp s;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = addc(a.v[i], b.v[i], c);
        return reduce_once(s);   // p < 2^254 so a+b < 2^255: no carry out of limb 7
    }
    ZK_HD static Fp sub(const Fp &a, const Fp &b) {
        Fp d;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb(a.v[i], b.v[i], bw);
        Fp r;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = addc(d.v[i], bw ? PR::P[i] : 0u, c);
        return r;
    }
    ZK_HD static Fp neg(const Fp &a) {
        if (a.is_zero()) return a;
        Fp r;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = subb(PR::P[i], a.v[i], bw);
        return r;
    }
    ZK_HD static Fp dbl(const Fp &a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs (== E.fr.mul / E.f1.mul).
    // Each row x*y[8] is formed as two interleaved sets of independent 64-bit MADs
    // (even columns E, odd columns O: no MAD depends on another's high word), then folded
    // into the accumulator with two carry chains.
    ZK_HD static void mad_row(u32 t[10], const u32 a[8], u32 bi) {
        u64 e0 = (u64)a[0] * bi, e2 = (u64)a[2] * bi, e4 = (u64)a[4] * bi, e6 = (u64)a[6] * bi;
        u64 o1 = (u64)a[1] * bi, o3 = (u64)a[3] * bi, o5 = (u64)a[5] * bi, o7 = (u64)a[7] * bi;
        u32 c = 0;
        t[0] = addc(t[0], (u32)e0, c);
        t[1] = addc(t[1], (u32)(e0 >> 32), c);
        t[2] = addc(t[2], (u32)e2, c);
        t[3] = addc(t[3], (u32)(e2 >> 32), c);
        t[4] = addc(t[4], (u32)e4, c);
        t[5] = addc(t[5], (u32)(e4 >> 32), c);
        t[6] = addc(t[6], (u32)e6, c);
        t[7] = addc(t[7], (u32)(e6 >> 32), c);
        t[8] = addc(t[8], 0u, c);
        c = 0;
        t[1] = addc(t[1], (u32)o1, c);
        t[2] = addc(t[2], (u32)(o1 >> 32), c);
        t[3] = addc(t[3], (u32)o3, c);
        t[4] = addc(t[4], (u32)(o3 >> 32), c);
        t[5] = addc(t[5], (u32)o5, c);
        t[6] = addc(t[6], (u32)(o5 >> 32), c);
        t[7] = addc(t[7], (u32)o7, c);
        t[8] = addc(t[8], (u32)(o7 >> 32), c);
    }
    ZK_HD static Fp mul(const Fp &a, const Fp &b) {
        u32 t[10];
#pragma unroll
        for (int i = 0; i < 10; i++) t[i] = 0;
        u32 p[8];
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = PR::P[i];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            mad_row(t, a.v, b.v[i]);              // t += a * b[i]      (t < 2p*2^32: 9 limbs)
            u32 m = t[0] * PR::N0INV;
            mad_row(t, p, m);                     // t += m * p         (low limb becomes 0)
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = t[j + 1];   // t >>= 32 (register renaming only)
            t[8] = 0;
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = t[i];
        return reduce_once(r);
    }
    ZK_HD static Fp sqr(const Fp &a) { return mul(a, a); }
    static constexpr bool FUSED_MULADD = false;

    ZK_HD static Fp to_mont(const Fp &a) { return mul(a, r2()); }
    ZK_HD static Fp from_mont(const Fp &a) {
        Fp o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    // a^(p-2): host-side only use (final affine conversion; 3 per proof)
    ZK_HD static Fp inv(const Fp &a) {
        Fp result = one();
        Fp base = a;
        // exponent p-2, little-endian bits
        u32 e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = PR::P[i];
        e[0] -= 2;   // P[0] >= 2 for both primes
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1u) result = mul(result, base);
            base = sqr(base);
        }
        return result;
    }
};

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

}   // namespace zk
