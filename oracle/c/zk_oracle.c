/* zk_oracle.c — CPU restatement of rapidsnark's Groth16 prove() path.  TEST INFRASTRUCTURE.
 *
 * ORACLE / CPU BASELINE ONLY: may be used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg — never by the product path (rapidsnark-old_amd/), which has no CPU fallback.
 *
 * Follows the reference's algorithm and stage order:
 *   prove()      src/groth16.cpp:48-254   (OpenMP `parallel for` at the same seven sites
 *                :56,:66,:89,:107,:125,:144,:158, including the 1024 striped omp locks of :63-85)
 *   file layout  src/zkey_utils.cpp:17-52, src/groth16.hpp:27-35 (44-byte packed Coef)
 * The arithmetic the reference takes from its ABSENT `depends/ffiasm` submodule
 * (.gitmodules:7-9, no pinned commit recoverable) is restated from its published algorithm:
 * 4x64-bit Montgomery (R = 2^256), radix-2 bit-reversal FFT, Pippenger multiexp with
 * per-task bucket arrays (window x point-slice), window c = clamp(log2(n/2), 2, 16) as BASELINE.md states, Horner over windows.
 * It is "a restatement of rapidsnark's CPU algorithm", NOT ffiasm: hand-written ADX assembly
 * may be 1.3-2x faster than this compiler-generated code (say so next to any timing).
 *
 * PARITY UNPINNED at the reference boundary (the reference ships no tests/vectors,
 * package.json:7).  This file is pinned against oracle/bn254.py + the trapdoor check via
 * tests/test_oracle_c.py on the committed golden fixtures.
 *
 * Curve arithmetic here is Jacobian (X,Y,Z) — deliberately different formulas from the
 * GPU path's XYZZ so the two are independent implementations.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
static int omp_get_num_threads(void) { return 1; }
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;     /* field element, little-endian limbs */

typedef struct {
    uint64_t p[4], r1[4], r2[4], ninv;
} field_t;

static const field_t FR = {
    {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full},
    {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull},
    0xc2e1f593efffffffull};
static const field_t FQ = {
    {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full},
    {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full},
    0x87d20782e4866389ull};

/* ------------------------------------------------------------------ prime field */
static inline int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, 32) == 0; }
static inline int geq(const uint64_t a[4], const uint64_t p[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > p[i]) return 1;
        if (a[i] < p[i]) return 0;
    }
    return 1;
}
static inline void sub_p(uint64_t a[4], const uint64_t p[4]) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - p[i] - bw;
        a[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
}
static inline void f_add(const field_t *F, fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->v[i] + b->v[i];
        t.v[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq(t.v, F->p)) sub_p(t.v, F->p);
    *r = t;
}
static inline void f_sub(const field_t *F, fe *r, const fe *a, const fe *b) {
    u128 bw = 0;
    fe t;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - bw;
        t.v[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t.v[i] + F->p[i];
            t.v[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *r = t;
}
static inline void f_neg(const field_t *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    fe z = {{0, 0, 0, 0}};
    f_sub(F, r, &z, a);
}
/* Montgomery product a*b*R^-1 mod p  (E.fr.mul / E.f1.mul).  Two builds of the SAME function (identical results, checked
 * against each other by tests/test_oracle_c.py):
 *   default          portable C: CIOS over unsigned __int128 (what gcc makes of it: mulx + one adc chain)
 *   -DZK_ORACLE_ADX  x86-64 with BMI2 + ADX: mulx with the two independent carry chains of adcx / adox — the instruction
 *                    mix of the "intel assembly with ADX extensions" the reference's README.md:67-69 names for its field
 *                    arithmetic (ffiasm generates it with nasm; ffiasm itself is absent from the checkout, so this is the
 *                    published technique restated, not its code).  BN254's primes leave the top bit of the top limb clear,
 *                    so the running value never needs a sixth limb ("no-carry" CIOS). */
#if defined(ZK_ORACLE_ADX) && defined(__x86_64__) && defined(__ADX__) && defined(__BMI2__)
#define ZK_ORACLE_VARIANT "adx"
/* one CIOS round: t += a * b[i];  m = t0 * ninv;  t += m * p;  t >>= 64.  T0..T3 hold t on entry (its fifth limb is zero),
 * T1, T2, T3, T4 on exit (T0 is scratch).  rdx, rax, r8, r9 are scratch. */
#define ZK_ADX_ROUND(I, T0, T1, T2, T3, T4)                                                                         \
    "movq " #I "*8(%[b]), %%rdx\n\t"                                                                                \
    "xorl %%eax, %%eax\n\t"                                                                                         \
    "mulxq 0(%[a]), %%rax, %%r8\n\t"   "adcxq %%rax, %[" #T0 "]\n\t"                                                \
    "mulxq 8(%[a]), %%rax, %%r9\n\t"   "adcxq %%rax, %[" #T1 "]\n\t"  "adoxq %%r8, %[" #T1 "]\n\t"                   \
    "mulxq 16(%[a]), %%rax, %%r8\n\t"  "adcxq %%rax, %[" #T2 "]\n\t"  "adoxq %%r9, %[" #T2 "]\n\t"                   \
    "mulxq 24(%[a]), %%rax, %[" #T4 "]\n\t" "adcxq %%rax, %[" #T3 "]\n\t"  "adoxq %%r8, %[" #T3 "]\n\t"              \
    "movl $0, %%eax\n\t"               "adcxq %%rax, %[" #T4 "]\n\t"  "adoxq %%rax, %[" #T4 "]\n\t"                  \
    "movq %[" #T0 "], %%rdx\n\t"       "imulq 96(%[F]), %%rdx\n\t"                                                  \
    "xorl %%eax, %%eax\n\t"                                                                                         \
    "mulxq 0(%[F]), %%rax, %%r8\n\t"   "adcxq %[" #T0 "], %%rax\n\t"                                                \
    "mulxq 8(%[F]), %%rax, %%r9\n\t"   "adcxq %%rax, %[" #T1 "]\n\t"  "adoxq %%r8, %[" #T1 "]\n\t"                   \
    "mulxq 16(%[F]), %%rax, %%r8\n\t"  "adcxq %%rax, %[" #T2 "]\n\t"  "adoxq %%r9, %[" #T2 "]\n\t"                   \
    "mulxq 24(%[F]), %%rax, %%r9\n\t"  "adcxq %%rax, %[" #T3 "]\n\t"  "adoxq %%r8, %[" #T3 "]\n\t"                   \
    "movl $0, %%eax\n\t"               "adcxq %%r9, %[" #T4 "]\n\t"   "adoxq %%rax, %[" #T4 "]\n\t"                  \
    "movl $0, %k[" #T0 "]\n\t"
static inline void f_mul(const field_t *F, fe *r, const fe *a, const fe *b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    __asm__(ZK_ADX_ROUND(0, t0, t1, t2, t3, t4)
            ZK_ADX_ROUND(1, t1, t2, t3, t4, t0)
            ZK_ADX_ROUND(2, t2, t3, t4, t0, t1)
            ZK_ADX_ROUND(3, t3, t4, t0, t1, t2)
            : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4)
            : [a] "r"(a->v), [b] "r"(b->v), [F] "r"(F), "m"(*a), "m"(*b), "m"(*F)
            : "rax", "rdx", "r8", "r9", "cc");
    /* four rounds rotate the names by four: the value is (t4, t0, t1, t2) */
    fe o = {{t4, t0, t1, t2}};
    if (geq(o.v, F->p)) sub_p(o.v, F->p);
    *r = o;
}
#else
#define ZK_ORACLE_VARIANT "generic"
static inline void f_mul(const field_t *F, fe *r, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[j] * b->v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->ninv;
        c = (u128)m * F->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F->p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o;
    memcpy(o.v, t, 32);
    if (t[4] || geq(o.v, F->p)) sub_p(o.v, F->p);
    *r = o;
}
#endif
static inline void f_sqr(const field_t *F, fe *r, const fe *a) { f_mul(F, r, a, a); }
static inline void f_one(const field_t *F, fe *r) { memcpy(r->v, F->r1, 32); }
static inline void f_to_mont(const field_t *F, fe *r, const fe *a) {
    fe r2;
    memcpy(r2.v, F->r2, 32);
    f_mul(F, r, a, &r2);
}
static inline void f_from_mont(const field_t *F, fe *r, const fe *a) {
    fe one = {{1, 0, 0, 0}};
    f_mul(F, r, a, &one);
}
static void f_pow(const field_t *F, fe *r, const fe *a, const uint64_t e[4]) {
    fe res, base = *a;
    f_one(F, &res);
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) f_mul(F, &res, &res, &base);
        f_sqr(F, &base, &base);
    }
    *r = res;
}
static void f_inv(const field_t *F, fe *r, const fe *a) {
    uint64_t e[4] = {F->p[0] - 2, F->p[1], F->p[2], F->p[3]};
    f_pow(F, r, a, e);
}

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1) */
typedef struct { fe a, b; } fe2;
static inline int f2_is_zero(const fe2 *x) { return fe_is_zero(&x->a) && fe_is_zero(&x->b); }
static inline int f2_eq(const fe2 *x, const fe2 *y) { return fe_eq(&x->a, &y->a) && fe_eq(&x->b, &y->b); }
static inline void f2_add(fe2 *r, const fe2 *x, const fe2 *y) { f_add(&FQ, &r->a, &x->a, &y->a); f_add(&FQ, &r->b, &x->b, &y->b); }
static inline void f2_sub(fe2 *r, const fe2 *x, const fe2 *y) { f_sub(&FQ, &r->a, &x->a, &y->a); f_sub(&FQ, &r->b, &x->b, &y->b); }
static inline void f2_neg(fe2 *r, const fe2 *x) { f_neg(&FQ, &r->a, &x->a); f_neg(&FQ, &r->b, &x->b); }
/* (a + b u)(c + d u), u^2 = -1: three base-field products (Karatsuba), as the reference's F2Field does (ffiasm f2field:
 * aA = a*c, bB = b*d, (a + b)(c + d) - aA - bB); the square is the "complex" one, two products */
static inline void f2_mul(fe2 *r, const fe2 *x, const fe2 *y) {
    fe aa, bb, s, t, m;
    f_mul(&FQ, &aa, &x->a, &y->a);
    f_mul(&FQ, &bb, &x->b, &y->b);
    f_add(&FQ, &s, &x->a, &x->b);
    f_add(&FQ, &t, &y->a, &y->b);
    f_mul(&FQ, &m, &s, &t);
    f_sub(&FQ, &m, &m, &aa);
    f_sub(&FQ, &r->b, &m, &bb);
    f_sub(&FQ, &r->a, &aa, &bb);
}
static inline void f2_sqr(fe2 *r, const fe2 *x) {
    fe s, d, ab;
    f_add(&FQ, &s, &x->a, &x->b);
    f_sub(&FQ, &d, &x->a, &x->b);
    f_mul(&FQ, &ab, &x->a, &x->b);
    f_mul(&FQ, &r->a, &s, &d);
    f_add(&FQ, &r->b, &ab, &ab);
}
static inline void f2_one(fe2 *r) { f_one(&FQ, &r->a); memset(&r->b, 0, 32); }
static void f2_inv(fe2 *r, const fe2 *x) {
    fe t0, t1, d;
    f_sqr(&FQ, &t0, &x->a);
    f_sqr(&FQ, &t1, &x->b);
    f_add(&FQ, &t0, &t0, &t1);
    f_inv(&FQ, &d, &t0);
    f_mul(&FQ, &r->a, &x->a, &d);
    f_mul(&FQ, &t1, &x->b, &d);
    f_neg(&FQ, &r->b, &t1);
}

/* ------------------------------------------------------------------ curves (Jacobian), generated twice */
#define DEFINE_CURVE(PFX, FE, F_IS_ZERO, F_EQ, F_ADD, F_SUB, F_NEG, F_MUL, F_SQR, F_ONE, F_INV)                        \
    typedef struct { FE x, y; } PFX##_aff;                                                                             \
    typedef struct { FE x, y, z; } PFX##_jac;                                                                          \
    static inline int PFX##_aff_is_inf(const PFX##_aff *p) { return F_IS_ZERO(&p->x) && F_IS_ZERO(&p->y); }            \
    static inline int PFX##_is_inf(const PFX##_jac *p) { return F_IS_ZERO(&p->z); }                                    \
    static inline void PFX##_set_inf(PFX##_jac *p) { memset(p, 0, sizeof *p); }                                        \
    static void PFX##_dbl(PFX##_jac *r, const PFX##_jac *p) {                                                          \
        if (PFX##_is_inf(p)) { *r = *p; return; }                                                                      \
        FE A, B, C, D, E, Fv, t, X3, Y3, Z3;                                                                           \
        F_SQR(&A, &p->x); F_SQR(&B, &p->y); F_SQR(&C, &B);                                                             \
        F_ADD(&t, &p->x, &B); F_SQR(&t, &t); F_SUB(&t, &t, &A); F_SUB(&t, &t, &C); F_ADD(&D, &t, &t);                  \
        F_ADD(&E, &A, &A); F_ADD(&E, &E, &A);                                                                          \
        F_SQR(&Fv, &E);                                                                                                \
        F_SUB(&X3, &Fv, &D); F_SUB(&X3, &X3, &D);                                                                      \
        F_SUB(&t, &D, &X3); F_MUL(&Y3, &E, &t);                                                                        \
        F_ADD(&C, &C, &C); F_ADD(&C, &C, &C); F_ADD(&C, &C, &C); F_SUB(&Y3, &Y3, &C);                                  \
        F_MUL(&Z3, &p->y, &p->z); F_ADD(&Z3, &Z3, &Z3);                                                                \
        r->x = X3; r->y = Y3; r->z = Z3;                                                                               \
    }                                                                                                                  \
    /* r = p + q (q affine) */                                                                                         \
    static void PFX##_madd(PFX##_jac *r, const PFX##_jac *p, const PFX##_aff *q) {                                     \
        if (PFX##_aff_is_inf(q)) { *r = *p; return; }                                                                  \
        if (PFX##_is_inf(p)) { r->x = q->x; r->y = q->y; F_ONE(&r->z); return; }                                       \
        FE Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;                                                            \
        F_SQR(&Z1Z1, &p->z); F_MUL(&U2, &q->x, &Z1Z1);                                                                 \
        F_MUL(&S2, &q->y, &p->z); F_MUL(&S2, &S2, &Z1Z1);                                                              \
        F_SUB(&H, &U2, &p->x); F_SUB(&rr, &S2, &p->y);                                                                 \
        if (F_IS_ZERO(&H)) {                                                                                           \
            if (F_IS_ZERO(&rr)) { PFX##_dbl(r, p); return; }                                                           \
            PFX##_set_inf(r); return;                                                                                  \
        }                                                                                                              \
        F_SQR(&HH, &H); F_MUL(&J, &H, &HH); F_MUL(&V, &p->x, &HH);                                                     \
        (void)I;                                                                                                       \
        F_SQR(&X3, &rr); F_SUB(&X3, &X3, &J); F_SUB(&X3, &X3, &V); F_SUB(&X3, &X3, &V);                                \
        F_SUB(&t, &V, &X3); F_MUL(&Y3, &rr, &t); F_MUL(&t, &p->y, &J); F_SUB(&Y3, &Y3, &t);                            \
        F_MUL(&Z3, &p->z, &H);                                                                                         \
        r->x = X3; r->y = Y3; r->z = Z3;                                                                               \
    }                                                                                                                  \
    static void PFX##_add(PFX##_jac *r, const PFX##_jac *p, const PFX##_jac *q) {                                      \
        if (PFX##_is_inf(q)) { *r = *p; return; }                                                                      \
        if (PFX##_is_inf(p)) { *r = *q; return; }                                                                      \
        FE Z1Z1, Z2Z2, U1, U2, S1, S2, H, HH, HHH, rr, V, t, X3, Y3, Z3;                                               \
        F_SQR(&Z1Z1, &p->z); F_SQR(&Z2Z2, &q->z);                                                                      \
        F_MUL(&U1, &p->x, &Z2Z2); F_MUL(&U2, &q->x, &Z1Z1);                                                            \
        F_MUL(&S1, &p->y, &q->z); F_MUL(&S1, &S1, &Z2Z2);                                                              \
        F_MUL(&S2, &q->y, &p->z); F_MUL(&S2, &S2, &Z1Z1);                                                              \
        F_SUB(&H, &U2, &U1); F_SUB(&rr, &S2, &S1);                                                                     \
        if (F_IS_ZERO(&H)) {                                                                                           \
            if (F_IS_ZERO(&rr)) { PFX##_dbl(r, p); return; }                                                           \
            PFX##_set_inf(r); return;                                                                                  \
        }                                                                                                              \
        F_SQR(&HH, &H); F_MUL(&HHH, &H, &HH); F_MUL(&V, &U1, &HH);                                                     \
        F_SQR(&X3, &rr); F_SUB(&X3, &X3, &HHH); F_SUB(&X3, &X3, &V); F_SUB(&X3, &X3, &V);                              \
        F_SUB(&t, &V, &X3); F_MUL(&Y3, &rr, &t); F_MUL(&t, &S1, &HHH); F_SUB(&Y3, &Y3, &t);                            \
        F_MUL(&Z3, &p->z, &q->z); F_MUL(&Z3, &Z3, &H);                                                                 \
        r->x = X3; r->y = Y3; r->z = Z3;                                                                               \
    }                                                                                                                  \
    static void PFX##_neg(PFX##_jac *r, const PFX##_jac *p) { *r = *p; F_NEG(&r->y, &p->y); }                          \
    static void PFX##_from_aff(PFX##_jac *r, const PFX##_aff *p) {                                                     \
        if (PFX##_aff_is_inf(p)) { PFX##_set_inf(r); return; }                                                         \
        r->x = p->x; r->y = p->y; F_ONE(&r->z);                                                                        \
    }                                                                                                                  \
    static void PFX##_to_aff(PFX##_aff *r, const PFX##_jac *p) {                                                       \
        if (PFX##_is_inf(p)) { memset(r, 0, sizeof *r); return; }                                                      \
        FE zi, zi2, zi3;                                                                                               \
        F_INV(&zi, &p->z); F_SQR(&zi2, &zi); F_MUL(&zi3, &zi2, &zi);                                                   \
        F_MUL(&r->x, &p->x, &zi2); F_MUL(&r->y, &p->y, &zi3);                                                          \
    }                                                                                                                  \
    /* scalar: 32 LE bytes standard form (E.g1.mulByScalar) */                                                         \
    static void PFX##_mul_scalar(PFX##_jac *r, const PFX##_jac *p, const uint8_t k[32]) {                              \
        PFX##_jac acc; PFX##_set_inf(&acc);                                                                            \
        for (int i = 255; i >= 0; i--) {                                                                               \
            PFX##_dbl(&acc, &acc);                                                                                     \
            if ((k[i >> 3] >> (i & 7)) & 1) PFX##_add(&acc, &acc, p);                                                  \
        }                                                                                                              \
        *r = acc;                                                                                                      \
    }                                                                                                                  \
    /* Pippenger with per-thread bucket arrays (ffiasm ParallelMultiexp shape), scheduled for many    \
       cores: tasks = (window, slice of the points); every task fills its own 2^c bucket array,      \
       slices are folded per bucket (packThreads), windows reduced by chunked running sums, then      \
       Horner.  No per-window barriers.  scalars: n x 32 B LE standard form; zero digits and          \
       infinity bases are skipped. */                                                                 \
    static void PFX##_msm(PFX##_jac *out, const PFX##_aff *bases, const uint8_t *scalars, uint64_t n) {                \
        PFX##_set_inf(out);                                                                                            \
        if (n == 0) return;                                                                                            \
        int nt = omp_get_max_threads();                                                                                \
        int c, tpw;                                                                                                    \
        choose_plan(n, nt, &c, &tpw);                                                                                  \
        int W = (256 + c - 1) / c;                                                                                     \
        uint64_t nb = ((uint64_t)1 << c);                                                                              \
        int ntask = W * tpw;                                                                                           \
        PFX##_jac *buckets = (PFX##_jac *)malloc(sizeof(PFX##_jac) * nb * (size_t)ntask);                              \
        PFX##_jac *wsum = (PFX##_jac *)malloc(sizeof(PFX##_jac) * (size_t)W);                                          \
        _Pragma("omp parallel for schedule(dynamic, 1)")                                                               \
        for (int t = 0; t < ntask; t++) {                                                                              \
            int w = t / tpw, j = t % tpw;                                                                              \
            PFX##_jac *B = buckets + (size_t)t * nb;                                                                   \
            memset(B, 0, sizeof(PFX##_jac) * nb);                                                                      \
            uint64_t lo = n * (uint64_t)j / tpw, hi = n * (uint64_t)(j + 1) / tpw;                                     \
            for (uint64_t i = lo; i < hi; i++) {                                                                       \
                /* the bucket of the point eight ahead is on its way while this one is added: a task's 2^c buckets (6 MB   \
                   of G1 Jacobian points at c = 16) do not stay in a core's cache, and without this the loop waits for    \
                   memory, not for the field arithmetic (no change of results; what any tuned CPU Pippenger does) */     \
                if (i + 8 < hi) {                                                                                      \
                    uint32_t dn = get_digit(scalars + (i + 8) * 32, w, c);                                             \
                    __builtin_prefetch(&B[dn], 1, 1);                                                                  \
                    __builtin_prefetch(&bases[i + 8], 0, 0);                                                           \
                }                                                                                                      \
                uint32_t d = get_digit(scalars + i * 32, w, c);                                                        \
                if (d) PFX##_madd(&B[d], &B[d], &bases[i]);                                                            \
            }                                                                                                          \
        }                                                                                                              \
        /* packThreads: fold slices 1.. into slice 0, per (window, bucket) */                                          \
        if (tpw > 1) {                                                                                                 \
            _Pragma("omp parallel for schedule(static) collapse(2)")                                                   \
            for (int w = 0; w < W; w++)                                                                                \
                for (uint64_t d = 1; d < nb; d++) {                                                                    \
                    PFX##_jac *B0 = buckets + (size_t)(w * tpw) * nb;                                                  \
                    for (int j = 1; j < tpw; j++) PFX##_add(&B0[d], &B0[d], &buckets[(size_t)(w * tpw + j) * nb + d]); \
                }                                                                                                      \
        }                                                                                                              \
        /* reduce: sum_d d*B[d] by running sums, chunked (parts per window) */                                         \
        {                                                                                                              \
            int parts = (nt + W - 1) / W;                                                                              \
            if ((uint64_t)parts > nb / 4) parts = (int)(nb / 4 ? nb / 4 : 1);                                          \
            PFX##_jac *psum = (PFX##_jac *)malloc(sizeof(PFX##_jac) * (size_t)parts * W);                              \
            _Pragma("omp parallel for schedule(dynamic, 1) collapse(2)")                                               \
            for (int w = 0; w < W; w++)                                                                                \
                for (int k = 0; k < parts; k++) {                                                                      \
                    const PFX##_jac *B0 = buckets + (size_t)(w * tpw) * nb;                                            \
                    uint64_t lo = 1 + (nb - 1) * (uint64_t)k / parts, hi = 1 + (nb - 1) * (uint64_t)(k + 1) / parts;   \
                    PFX##_jac run, sum; PFX##_set_inf(&run); PFX##_set_inf(&sum);                                      \
                    for (uint64_t d = hi; d-- > lo;) { PFX##_add(&run, &run, &B0[d]); PFX##_add(&sum, &sum, &run); }   \
                    /* sum = sum_{d in [lo,hi)} (d-lo+1) B[d]; add (lo-1)*run */                                       \
                    uint8_t kk[32] = {0}; uint64_t m = lo - 1; memcpy(kk, &m, 8);                                      \
                    PFX##_jac mr; PFX##_mul_scalar(&mr, &run, kk); PFX##_add(&sum, &sum, &mr);                         \
                    psum[w * parts + k] = sum;                                                                         \
                }                                                                                                      \
            for (int w = 0; w < W; w++) {                                                                              \
                PFX##_jac tot; PFX##_set_inf(&tot);                                                                    \
                for (int k = 0; k < parts; k++) PFX##_add(&tot, &tot, &psum[w * parts + k]);                           \
                wsum[w] = tot;                                                                                         \
            }                                                                                                          \
            free(psum);                                                                                                \
        }                                                                                                              \
        PFX##_jac acc; PFX##_set_inf(&acc);                                                                            \
        for (int w = W - 1; w >= 0; w--) {                                                                             \
            for (int k = 0; k < c; k++) PFX##_dbl(&acc, &acc);                                                         \
            PFX##_add(&acc, &acc, &wsum[w]);                                                                           \
        }                                                                                                              \
        *out = acc;                                                                                                    \
        free(buckets); free(wsum);                                                                                     \
    }

/* window bits: c = clamp(floor(log2(n / 2)), 2, 16) — the rule BASELINE.md §2 states for the reference's
 * Pippenger (ffiasm clamps to [2,16]); only the number of point slices per window (how the nt threads are
 * spread over W windows) is chosen here.  ZK_ORACLE_WINDOW=cost restores the round-1 cost model over c
 * (identical choice, c = 16, from 2^17 points up). */
static void choose_plan(uint64_t n, int nt, int *c_out, int *tpw_out) {
    const char *mode = getenv("ZK_ORACLE_WINDOW");
    int c = 2;
    if (mode && !strcmp(mode, "cost")) {
        double best = 1e300;
        for (int cc = 2; cc <= 16; cc++) {
            int W = (256 + cc - 1) / cc;
            int tpw = nt / W;
            if (tpw < 1) tpw = 1;
            if ((uint64_t)tpw > n) tpw = (int)n;
            double nb = (double)((uint64_t)1 << cc);
            int par = W * tpw < nt ? W * tpw : nt;
            double cost = W * ((double)n + (tpw > 1 ? tpw * nb : 0.0) + 2.0 * nb) / par + 2.0 * nb / 4.0;
            if (cost < best) { best = cost; c = cc; }
        }
    } else {
        uint64_t h = n / 2;
        int lg = 0;
        while (h > 1) { h >>= 1; lg++; }
        c = lg < 2 ? 2 : (lg > 16 ? 16 : lg);
    }
    int W = (256 + c - 1) / c;
    int tpw = nt / W;
    if (tpw < 1) tpw = 1;
    if ((uint64_t)tpw > n) tpw = (int)(n ? n : 1);
    *c_out = c;
    *tpw_out = tpw;
}
/* unsigned c-bit digit w of a 256-bit LE scalar */
static inline uint32_t get_digit(const uint8_t *s, int w, int c) {
    int bit = w * c;
    if (bit >= 256) return 0;
    uint64_t buf = 0;
    int byte = bit >> 3;
    for (int k = 0; k < 8 && byte + k < 32; k++) buf |= (uint64_t)s[byte + k] << (8 * k);
    buf >>= (bit & 7);
    return (uint32_t)(buf & (((uint64_t)1 << c) - 1));
}

#define FQ_ADD(r, a, b) f_add(&FQ, r, a, b)
#define FQ_SUB(r, a, b) f_sub(&FQ, r, a, b)
#define FQ_NEG(r, a) f_neg(&FQ, r, a)
#define FQ_MUL(r, a, b) f_mul(&FQ, r, a, b)
#define FQ_SQR(r, a) f_sqr(&FQ, r, a)
#define FQ_ONE(r) f_one(&FQ, r)
#define FQ_INV(r, a) f_inv(&FQ, r, a)
DEFINE_CURVE(g1, fe, fe_is_zero, fe_eq, FQ_ADD, FQ_SUB, FQ_NEG, FQ_MUL, FQ_SQR, FQ_ONE, FQ_INV)
DEFINE_CURVE(g2, fe2, f2_is_zero, f2_eq, f2_add, f2_sub, f2_neg, f2_mul, f2_sqr, f2_one, f2_inv)

/* ------------------------------------------------------------------ FFT (ffiasm FFT<Fr> restated) */
/* w_{2^28} = 5^((r-1)/2^28), standard form */
static const fe ROOT_2_28 = {{0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull}};

static void fr_root(fe *r, int k) { /* primitive 2^k-th root, Montgomery */
    fe w;
    f_to_mont(&FR, &w, &ROOT_2_28);
    for (int i = k; i < 28; i++) f_sqr(&FR, &w, &w);
    *r = w;
}
static int ilog2(uint64_t n) {
    int l = 0;
    while (((uint64_t)1 << l) < n) l++;
    return l;
}
static uint32_t brev32(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
/* in place, natural order in and out, Montgomery elements; inverse includes 1/n */
static void fr_fft(fe *a, uint64_t n, int inverse) {
    int logn = ilog2(n);
    if (n <= 1) return;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) {
        uint64_t j = brev32((uint32_t)i, logn);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    fe wn;
    fr_root(&wn, logn);
    if (inverse) f_inv(&FR, &wn, &wn);
    /* root table w_n^k, k < n/2 (the reference precomputes a table too, src/groth16.hpp:94) */
    fe *tw = (fe *)malloc(sizeof(fe) * (n / 2));
    {
        int nt = omp_get_max_threads();
        uint64_t half = n / 2;
#pragma omp parallel for schedule(static)
        for (int t = 0; t < nt; t++) {
            uint64_t lo = half * (uint64_t)t / nt, hi = half * (uint64_t)(t + 1) / nt;
            if (lo >= hi) continue;
            uint64_t e[4] = {lo, 0, 0, 0};
            fe cur;
            f_pow(&FR, &cur, &wn, e);
            for (uint64_t k = lo; k < hi; k++) { tw[k] = cur; f_mul(&FR, &cur, &cur, &wn); }
        }
    }
    for (uint64_t m = 1; m < n; m <<= 1) {
        uint64_t stride = n / (2 * m);
#pragma omp parallel for schedule(static)
        for (uint64_t b = 0; b < n / 2; b++) {
            uint64_t k = (b / m) * 2 * m, j = b % m;
            fe u = a[k + j], v;
            f_mul(&FR, &v, &a[k + j + m], &tw[j * stride]);
            f_add(&FR, &a[k + j], &u, &v);
            f_sub(&FR, &a[k + j + m], &u, &v);
        }
    }
    if (inverse) {
        fe nn = {{n, 0, 0, 0}}, ninv;
        f_to_mont(&FR, &nn, &nn);
        f_inv(&FR, &ninv, &nn);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < n; i++) f_mul(&FR, &a[i], &a[i], &ninv);
    }
    free(tw);
}

/* ------------------------------------------------------------------ prove() (src/groth16.cpp:48-254) */
#pragma pack(push, 1)
typedef struct { uint32_t m, c, s; fe coef; } coef_t; /* src/groth16.hpp:27-35 */
#pragma pack(pop)

typedef struct {
    uint32_t nVars, nPublic, domainSize;
    uint64_t nCoefs;
    const void *vk_alpha1, *vk_beta1, *vk_beta2, *vk_delta1, *vk_delta2;
    const void *coefs, *pointsA, *pointsB1, *pointsB2, *pointsC, *pointsH;
} oracle_zkey_view;

typedef struct { uint8_t pih[64], pi_a[64], pib1[64], pi_b[128], pi_c[64]; } oracle_msm_sums;

/* steps 1-5: h[] standard form (src/groth16.cpp:52-163) */
static fe *compute_h(const oracle_zkey_view *z, const fe *wtns) {
    uint64_t n = z->domainSize;
    fe *a = (fe *)calloc(n, sizeof(fe)), *b = (fe *)calloc(n, sizeof(fe)), *c = (fe *)malloc(n * sizeof(fe));
    const coef_t *coefs = (const coef_t *)((const uint8_t *)z->coefs + 4); /* :38 */
    /* :62-85 — exactly the reference's scheme: omp parallel for over the records, the Montgomery
       product outside the lock, the accumulation under one of NLOCKS = 1024 locks striped by c */
#ifdef _OPENMP
    enum { NLOCKS = 1024 };
    omp_lock_t locks[NLOCKS];
    for (int i = 0; i < NLOCKS; i++) omp_init_lock(&locks[i]);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < z->nCoefs; i++) {
        coef_t rec;
        memcpy(&rec, &coefs[i], sizeof rec);
        fe *ab = rec.m == 0 ? a : b;
        fe aux;
        f_mul(&FR, &aux, &wtns[rec.s], &rec.coef);
        omp_set_lock(&locks[rec.c % NLOCKS]);
        f_add(&FR, &ab[rec.c], &ab[rec.c], &aux);
        omp_unset_lock(&locks[rec.c % NLOCKS]);
    }
    for (int i = 0; i < NLOCKS; i++) omp_destroy_lock(&locks[i]);
#else
    for (uint64_t i = 0; i < z->nCoefs; i++) {
        coef_t rec;
        memcpy(&rec, &coefs[i], sizeof rec);
        fe *ab = rec.m == 0 ? a : b;
        fe aux;
        f_mul(&FR, &aux, &wtns[rec.s], &rec.coef);
        f_add(&FR, &ab[rec.c], &ab[rec.c], &aux);
    }
#endif
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) f_mul(&FR, &c[i], &a[i], &b[i]); /* :89-96 */
    int domainPower = ilog2(n);
    fe w2n;
    fr_root(&w2n, domainPower + 1);
    fe *polys[3] = {a, b, c};
    for (int k = 0; k < 3; k++) { /* :101-155 */
        fe *x = polys[k];
        fr_fft(x, n, 1);
        int nt = omp_get_max_threads();
#pragma omp parallel for schedule(static)
        for (int t = 0; t < nt; t++) { /* x[i] *= w_2n^i  (fft->root(domainPower+1, i)) */
            uint64_t lo = n * (uint64_t)t / nt, hi = n * (uint64_t)(t + 1) / nt;
            if (lo >= hi) continue;
            uint64_t e[4] = {lo, 0, 0, 0};
            fe cur;
            f_pow(&FR, &cur, &w2n, e);
            for (uint64_t i = lo; i < hi; i++) { f_mul(&FR, &x[i], &x[i], &cur); f_mul(&FR, &cur, &cur, &w2n); }
        }
        fr_fft(x, n, 0);
    }
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) { /* :158-163 */
        f_mul(&FR, &a[i], &a[i], &b[i]);
        f_sub(&FR, &a[i], &a[i], &c[i]);
        f_from_mont(&FR, &a[i], &a[i]);
    }
    free(b);
    free(c);
    return a;
}

static void msm_sums(const oracle_zkey_view *z, const uint8_t *wtns, oracle_msm_sums *out, fe **h_out) {
    fe *h = compute_h(z, (const fe *)wtns);
    g1_jac pih, pi_a, pib1, pi_c;
    g2_jac pi_b;
    g1_msm(&pih, (const g1_aff *)z->pointsH, (const uint8_t *)h, z->domainSize);                       /* :173 */
    g1_msm(&pi_a, (const g1_aff *)z->pointsA, wtns, z->nVars);                                         /* :183 */
    g1_msm(&pib1, (const g1_aff *)z->pointsB1, wtns, z->nVars);                                        /* :190 */
    g2_msm(&pi_b, (const g2_aff *)z->pointsB2, wtns, z->nVars);                                        /* :197 */
    g1_msm(&pi_c, (const g1_aff *)z->pointsC, wtns + (size_t)(z->nPublic + 1) * 32, z->nVars - z->nPublic - 1); /* :204 */
    g1_to_aff((g1_aff *)out->pih, &pih);
    g1_to_aff((g1_aff *)out->pi_a, &pi_a);
    g1_to_aff((g1_aff *)out->pib1, &pib1);
    g2_to_aff((g2_aff *)out->pi_b, &pi_b);
    g1_to_aff((g1_aff *)out->pi_c, &pi_c);
    if (h_out) *h_out = h; else free(h);
}

static void final_assembly(const oracle_zkey_view *z, const oracle_msm_sums *m, const uint8_t r[32], const uint8_t s[32], uint8_t proof[256]) {
    g1_jac pi_a, pib1, pi_c, pih, p1, d1;
    g2_jac pi_b, p2, d2;
    g1_from_aff(&pi_a, (const g1_aff *)m->pi_a);
    g1_from_aff(&pib1, (const g1_aff *)m->pib1);
    g1_from_aff(&pi_c, (const g1_aff *)m->pi_c);
    g1_from_aff(&pih, (const g1_aff *)m->pih);
    g2_from_aff(&pi_b, (const g2_aff *)m->pi_b);
    g1_from_aff(&d1, (const g1_aff *)z->vk_delta1);
    g2_from_aff(&d2, (const g2_aff *)z->vk_delta2);
    g1_madd(&pi_a, &pi_a, (const g1_aff *)z->vk_alpha1); /* :222 */
    g1_mul_scalar(&p1, &d1, r); g1_add(&pi_a, &pi_a, &p1); /* :223-224 */
    g2_madd(&pi_b, &pi_b, (const g2_aff *)z->vk_beta2); /* :226 */
    g2_mul_scalar(&p2, &d2, s); g2_add(&pi_b, &pi_b, &p2); /* :227-228 */
    g1_madd(&pib1, &pib1, (const g1_aff *)z->vk_beta1); /* :230 */
    g1_mul_scalar(&p1, &d1, s); g1_add(&pib1, &pib1, &p1); /* :231-232 */
    g1_add(&pi_c, &pi_c, &pih); /* :234 */
    g1_mul_scalar(&p1, &pi_a, s); g1_add(&pi_c, &pi_c, &p1); /* :236-237 */
    g1_mul_scalar(&p1, &pib1, r); g1_add(&pi_c, &pi_c, &p1); /* :239-240 */
    fe fr_, fs_, rs;
    memcpy(&fr_, r, 32); memcpy(&fs_, s, 32);
    f_mul(&FR, &rs, &fr_, &fs_); f_to_mont(&FR, &rs, &rs); /* :242-243 */
    g1_mul_scalar(&p1, &d1, (const uint8_t *)&rs); g1_neg(&p1, &p1); g1_add(&pi_c, &pi_c, &p1); /* :245-246 */
    g1_to_aff((g1_aff *)proof, &pi_a); /* :249-251 */
    g2_to_aff((g2_aff *)(proof + 64), &pi_b);
    g1_to_aff((g1_aff *)(proof + 192), &pi_c);
}

/* ------------------------------------------------------------------ exported C API (ctypes) */
int oracle_num_threads(void) { return omp_get_max_threads(); }
const char *oracle_variant(void) { return ZK_ORACLE_VARIANT; }      /* "adx" or "generic": which build of f_mul this library holds */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void oracle_fr_mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) f_mul(&FR, (fe *)out + i, (const fe *)a + i, (const fe *)b + i);
}
void oracle_fq_mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) f_mul(&FQ, (fe *)out + i, (const fe *)a + i, (const fe *)b + i);
}
void oracle_fr_fft(uint8_t *data, uint64_t n, int inverse) { fr_fft((fe *)data, n, inverse); }

void oracle_msm_g1(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, uint64_t n) {
    g1_jac r;
    g1_msm(&r, (const g1_aff *)bases, scalars, n);
    g1_to_aff((g1_aff *)out, &r);
}
void oracle_msm_g2(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, uint64_t n) {
    g2_jac r;
    g2_msm(&r, (const g2_aff *)bases, scalars, n);
    g2_to_aff((g2_aff *)out, &r);
}
/* h (n x 32 B standard form) from a zkey view + witness */
void oracle_compute_h(const oracle_zkey_view *z, const uint8_t *wtns, uint8_t *h_out) {
    fe *h = compute_h(z, (const fe *)wtns);
    memcpy(h_out, h, (size_t)z->domainSize * 32);
    free(h);
}
void oracle_prove_msm(const oracle_zkey_view *z, const uint8_t *wtns, oracle_msm_sums *out) { msm_sums(z, wtns, out, NULL); }
void oracle_prove(const oracle_zkey_view *z, const uint8_t *wtns, const uint8_t r[32], const uint8_t s[32], uint8_t proof[256]) {
    oracle_msm_sums m;
    msm_sums(z, wtns, &m, NULL);
    final_assembly(z, &m, r, s, proof);
}

/* ---- synthetic tables (SURVEY §8d "perf-only point tables"): P_i = (k0 + i*kq) * G, affine
 * Montgomery, generated by an additive chain + batch inversion.  Known discrete logs make
 * full-size MSM results checkable in Fr alone. */
#define DEFINE_CHAIN(PFX, FE, F_MUL, F_INV, F_SQR, F_ONE)                                                              \
    /* out[i] = P0 + i*Q for affine points P0, Q */                                                                    \
    void oracle_chainp_##PFX(uint8_t *out, uint64_t n, const uint8_t p0[sizeof(PFX##_aff)], const uint8_t q[sizeof(PFX##_aff)]) { \
        if (!n) return;                                                                                                \
        PFX##_jac P0, Q;                                                                                               \
        PFX##_from_aff(&P0, (const PFX##_aff *)p0);                                                                    \
        PFX##_from_aff(&Q, (const PFX##_aff *)q);                                                                      \
        PFX##_aff Qa = *(const PFX##_aff *)q;                                                                          \
        int nt = omp_get_max_threads();                                                                                \
        if ((uint64_t)nt > n) nt = (int)n;                                                                             \
        _Pragma("omp parallel for schedule(static) num_threads(nt)")                                                   \
        for (int t = 0; t < nt; t++) {                                                                                 \
            uint64_t lo = n * (uint64_t)t / nt, hi = n * (uint64_t)(t + 1) / nt;                                       \
            if (lo >= hi) continue;                                                                                    \
            /* start = P0 + lo*Q */                                                                                    \
            PFX##_jac P, T; uint8_t kk[32] = {0}; memcpy(kk, &lo, 8);                                                  \
            PFX##_mul_scalar(&T, &Q, kk); PFX##_add(&P, &P0, &T);                                                      \
            uint64_t cnt = hi - lo;                                                                                    \
            PFX##_jac *js = (PFX##_jac *)malloc(sizeof(PFX##_jac) * cnt);                                              \
            FE *pref = (FE *)malloc(sizeof(FE) * cnt);                                                                 \
            for (uint64_t i = 0; i < cnt; i++) { js[i] = P; PFX##_madd(&P, &P, &Qa); }                                 \
            /* batch inversion of z (none is zero: the chain never hits infinity for these k) */                       \
            FE acc; F_ONE(&acc);                                                                                       \
            for (uint64_t i = 0; i < cnt; i++) { pref[i] = acc; F_MUL(&acc, &acc, &js[i].z); }                         \
            FE inv; F_INV(&inv, &acc);                                                                                 \
            for (uint64_t i = cnt; i-- > 0;) {                                                                         \
                FE zi, zi2, zi3; F_MUL(&zi, &inv, &pref[i]); F_MUL(&inv, &inv, &js[i].z);                              \
                F_SQR(&zi2, &zi); F_MUL(&zi3, &zi2, &zi);                                                              \
                PFX##_aff *o = (PFX##_aff *)out + lo + i;                                                              \
                F_MUL(&o->x, &js[i].x, &zi2); F_MUL(&o->y, &js[i].y, &zi3);                                            \
            }                                                                                                          \
            free(js); free(pref);                                                                                      \
        }                                                                                                              \
    }                                                                                                                  \
    /* out[i] = (k0 + i*kq) * gen */                                                                                   \
    void oracle_chain_##PFX(uint8_t *out, uint64_t n, const uint8_t gen[sizeof(PFX##_aff)], const uint8_t k0[32], const uint8_t kq[32]) { \
        PFX##_jac G, A, B; PFX##_aff a, b;                                                                             \
        PFX##_from_aff(&G, (const PFX##_aff *)gen);                                                                    \
        PFX##_mul_scalar(&A, &G, k0); PFX##_mul_scalar(&B, &G, kq);                                                    \
        PFX##_to_aff(&a, &A); PFX##_to_aff(&b, &B);                                                                    \
        oracle_chainp_##PFX(out, n, (const uint8_t *)&a, (const uint8_t *)&b);                                         \
    }
DEFINE_CHAIN(g1, fe, FQ_MUL, FQ_INV, FQ_SQR, FQ_ONE)
DEFINE_CHAIN(g2, fe2, f2_mul, f2_inv, f2_sqr, f2_one)

/* k*P for a single affine point (used by tests to turn a known discrete log into the expected point) */
void oracle_g1_mul(uint8_t out[64], const uint8_t p[64], const uint8_t k[32]) {
    g1_jac P, R;
    g1_from_aff(&P, (const g1_aff *)p);
    g1_mul_scalar(&R, &P, k);
    g1_to_aff((g1_aff *)out, &R);
}
void oracle_g2_mul(uint8_t out[128], const uint8_t p[128], const uint8_t k[32]) {
    g2_jac P, R;
    g2_from_aff(&P, (const g2_aff *)p);
    g2_mul_scalar(&R, &P, k);
    g2_to_aff((g2_aff *)out, &R);
}
