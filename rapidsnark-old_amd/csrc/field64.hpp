// Host-side BN254 field arithmetic, 4 x 64-bit limbs (x86-64, unsigned __int128).
//
// Used ONLY for the serial O(windows) tail of a proof — the window Horner combine and
// the final assembly of src/groth16.cpp:219-251 (6 scalar muls, ~10 adds, 3 inversions)
// — where one host core (~30 ns per Montgomery product) beats one GPU lane (~1 us) by 30x.
// Same byte layout as the device Fp (and as the reference's FrElement): memcpy-compatible.
#pragma once
#include <stdint.h>
#include <string.h>
#include "field.hpp"

namespace zk {

template <class PR>
struct Fp64 {
    uint64_t v[4];
    typedef unsigned __int128 u128;

    static uint64_t P(int i) { return (uint64_t)PR::P[2 * i] | ((uint64_t)PR::P[2 * i + 1] << 32); }
    static uint64_t n0inv() {
        // -p^-1 mod 2^64 by Newton iteration from the 32-bit constant
        uint64_t p0 = P(0);
        uint64_t x = (uint64_t)(0u - PR::N0INV);   // p^-1 mod 2^32 (approx seed)
        for (int i = 0; i < 4; i++) x *= 2 - p0 * x;
        return 0 - x;
    }
    static Fp64 zero() { Fp64 r; memset(r.v, 0, 32); return r; }
    static Fp64 one() { Fp64 r; memcpy(r.v, PR::R1, 32); return r; }
    static Fp64 r2() { Fp64 r; memcpy(r.v, PR::R2, 32); return r; }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
    bool is_zero_raw() const { return is_zero(); }
    bool operator==(const Fp64 &o) const { return memcmp(v, o.v, 32) == 0; }
    bool operator!=(const Fp64 &o) const { return !(*this == o); }

    static bool geq_p(const uint64_t a[4]) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] > P(i)) return true;
            if (a[i] < P(i)) return false;
        }
        return true;
    }
    static void sub_p(uint64_t a[4]) {
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a[i] - P(i) - bw;
            a[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
    }
    static Fp64 add(const Fp64 &a, const Fp64 &b) {
        Fp64 r;
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)a.v[i] + b.v[i];
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_p(r.v)) sub_p(r.v);
        return r;
    }
    static Fp64 sub(const Fp64 &a, const Fp64 &b) {
        Fp64 r;
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a.v[i] - b.v[i] - bw;
            r.v[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
        if (bw) {
            u128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (u128)r.v[i] + P(i);
                r.v[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    static Fp64 neg(const Fp64 &a) { return a.is_zero() ? a : sub(zero(), a); }
    static Fp64 dbl(const Fp64 &a) { return add(a, a); }
    static Fp64 mul(const Fp64 &a, const Fp64 &b) {
        static const uint64_t ninv = n0inv();
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a.v[j] * b.v[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * ninv;
            c = (u128)m * P(0) + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)m * P(j) + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fp64 r;
        memcpy(r.v, t, 32);
        if (t[4] || geq_p(r.v)) sub_p(r.v);
        return r;
    }
    static Fp64 sqr(const Fp64 &a) { return mul(a, a); }
    static constexpr bool FUSED_MULADD = false;
    static Fp64 to_mont(const Fp64 &a) { return mul(a, r2()); }
    static Fp64 from_mont(const Fp64 &a) {
        Fp64 o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    static Fp64 inv(const Fp64 &a) {   // a^(p-2)
        Fp64 result = one(), base = a;
        uint64_t e[4] = {P(0) - 2, P(1), P(2), P(3)};
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) result = mul(result, base);
            base = sqr(base);
        }
        return result;
    }
};

typedef Fp64<FrParams> Fr64;
typedef Fp64<FqParams> Fq64;

}   // namespace zk
