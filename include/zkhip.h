/* zkhip.h — C ABI of the MI355X-native Groth16 hot path (libzkhip.so).
 *
 * The reference (iden3/rapidsnark-old) has no FFI layer: its seam is the C++ template
 * `Groth16::Prover<Engine>` (reference src/groth16.hpp:37-121).  This header is the
 * C-ABI a maintainer would bind in its place; every entry point cites the reference
 * interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Byte conventions are the reference's own:
 *   Fr / Fq element : 32 bytes little-endian (== FrElement / 4 x u64)
 *   G1 affine       : x|y, Montgomery form (R = 2^256), 64 bytes; all-zero = infinity
 *   G2 affine       : x.a|x.b|y.a|y.b, Montgomery form, 128 bytes
 *   witness         : nVars x 32 B, standard (non-Montgomery) form   (src/main_prover.cpp:74)
 *
 * All functions return 0 on success, non-zero on error; zk_last_error() gives the
 * message for the calling thread.  Nothing throws across this boundary.  There is NO
 * CPU fallback: if no HIP device is usable every call fails with an error.
 */
#ifndef ZKHIP_H
#define ZKHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_prover zk_prover;

/* The 15 arguments of Groth16::makeProver<Engine>() (src/groth16.hpp:104-121,
 * call site src/main_prover.cpp:57-73), plus section byte sizes for bounds checks. */
typedef struct zk_zkey_view {
    uint32_t nVars;
    uint32_t nPublic;
    uint32_t domainSize;
    uint64_t nCoefs;
    const void *vk_alpha1;   /* G1, zkey section 2 */
    const void *vk_beta1;    /* G1 */
    const void *vk_beta2;    /* G2 */
    const void *vk_delta1;   /* G1 */
    const void *vk_delta2;   /* G2 */
    const void *coefs;       /* section 4 INCLUDING its leading u32 count (src/groth16.cpp:38 skips 4 bytes) */
    const void *pointsA;     /* section 5: nVars x G1 */
    const void *pointsB1;    /* section 6: nVars x G1 */
    const void *pointsB2;    /* section 7: nVars x G2 */
    const void *pointsC;     /* section 8: (nVars-nPublic-1) x G1 */
    const void *pointsH;     /* section 9: domainSize x G1 */
    uint64_t coefs_bytes, pointsA_bytes, pointsB1_bytes, pointsB2_bytes, pointsC_bytes, pointsH_bytes;
} zk_zkey_view;

typedef struct zk_opts {
    int32_t device;          /* HIP device ordinal; -1 = current device */
    uint32_t shard_index;    /* this prover holds shard `shard_index` of `shard_count` of every MSM   */
    uint32_t shard_count;    /*   point table (contiguous index slices); 0 or 1 = whole    */
    uint32_t window_bits;    /* Pippenger window c; 0 = choose from the size                           */
    uint32_t flags;          /* ZK_FLAG_* */
    uint32_t batch;          /* 0/1 = one witness per submission; 2..ZK_MAX_BATCH = up to that many witnesses of this
                              * circuit proved by ONE set of kernel launches (zk_prove_batch_*; small circuits, where a
                              * proof is bound by kernel latencies: DESIGN.md section 5).  Needs ZK_FLAG_PRECOMP, unsharded. */
} zk_opts;
#define ZK_MAX_BATCH 16

#define ZK_FLAG_TIMINGS 1u   /* record per-stage hipEvent timings (zk_prover_timings) */
#define ZK_FLAG_PARTITIONED_CHAIN 4u   /* sharded provers only (shard_count 2, 4 or 8): the A.w/B.w rows and the six
                              * transforms are PARTITIONED over the shards too (each holds the block of rows its H
                              * table slice covers) instead of replicated; such provers are driven through
                              * zk_multi_prove* (one process) or zk_shard_* (one process per GPU) */
#define ZK_FLAG_PRECOMP 2u   /* window-precomputed point tables: W x the table memory in HBM and a longer
                              * zk_prover_create, ~19 % fewer point additions per proof (same results) */

#define ZK_FLAG_PRECOMP_HALF 16u   /* (implies ZK_FLAG_PRECOMP) table rows for every SECOND window only: ceil(W/2) x the table memory
                              * instead of W x (7 instead of 13 at c = 20) and the same additions per point.  The digits of the odd
                              * windows add the neighbouring even window's row into a second bucket set, whose sum the host doubles
                              * c times: two bucket reductions per MSM instead of one.  For provers that hold several keys
                              * (proverServer, src/main_proofserver.cpp:12-26) and for circuits whose full tables do not fit (2^25
                              * on one MI355X).  Same proofs.  Not with opts.batch. */
#define ZK_FLAG_SPARSE_WITNESS 8u   /* with ZK_FLAG_PRECOMP: the four witness MSMs (A, B1, B2, C; src/groth16.cpp:180-204) use a
                              * 16-bit window (2^15 buckets per set) instead of the size-based one (2^19 at 2^22 constraints).
                              * For CIRCUIT witnesses — mostly 0, 1 and small values: few non-zero digits — the additions are few
                              * and the bucket reductions, which do not shrink with the witness, dominate those MSMs; with
                              * uniformly random scalars (the benchmark's worst case) it costs three more additions per point.
                              * Same proofs either way.  MSM H (scalars a.b - c: always full-size) keeps its window. */

/* Same bytes as Proof<Engine>{A,B,C} (src/groth16.hpp:13-24): affine, Montgomery LE. */
typedef struct zk_proof {
    uint8_t A[64];
    uint8_t B[128];
    uint8_t C[64];
} zk_proof;

/* The five multi-exponentiation results of src/groth16.cpp:171-204 before the final
 * assembly, as affine Montgomery points (all-zero = infinity).  With sharding these are
 * PARTIAL sums over this prover's slice; partial sums add up across shards. */
typedef struct zk_msm_sums {
    uint8_t pih[64];
    uint8_t pi_a[64];
    uint8_t pib1[64];
    uint8_t pi_b[128];
    uint8_t pi_c[64];
} zk_msm_sums;

const char *zk_last_error(void);
int zk_device_count(int *count);

/* makeProver (src/groth16.cpp:9-46) + Prover ctor (src/groth16.hpp:57-95).  One-off work:
 * CSR build of the coefficient records, upload of the point tables and twiddle tables.
 * The host image may be freed after this returns (the reference Prover borrows it). */
int zk_prover_create(zk_prover **out, const zk_zkey_view *zkey, const zk_opts *opts);
void zk_prover_destroy(zk_prover *p);

/* Prover::prove (src/groth16.cpp:48-254).  wtns: nVars x 32 B standard form (host memory).
 * r32/s32: 32-byte LE scalars replacing randombytes_buf (src/groth16.cpp:216-217); NULL
 * draws 31 random bytes each exactly as the reference does. */
int zk_prove(zk_prover *p, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out);
/* Same with the witness already resident in device memory (HBM) on the prover's device. */
int zk_prove_dev(zk_prover *p, const void *d_wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out);

/* Throughput mode: the same prove(), split so that consecutive proofs overlap.  The reference
 * proves strictly one at a time (src/fullprover.cpp:96-97: one worker thread); on the GPU the
 * latency-bound front of proof k+1 (digit sort, A.w/B.w, NTTs) hides under the tail of proof k
 * (bucket reductions, D2H, host Horner + final assembly, src/groth16.cpp:219-251).
 * zk_prove_dev_submit enqueues all device work of one proof and returns; at most ZK_MAX_IN_FLIGHT
 * proofs may be in flight per prover; per-proof buffers are allocated the first time a depth is reached.
 * Large circuits: two keep the chip busy when witnesses are resident in HBM, a third and fourth hide the upload of a
 * host witness (a proof cannot start before its witness has arrived).  Small circuits (below ~2^19) are
 * bound by the serial latency of their ~60 small kernels, not by throughput: there more proofs in flight
 * (on the prover's four lanes of streams) and batched submissions are what fills the GPU.  zk_prove_collect blocks until the OLDEST submitted proof is complete
 * and writes it.  d_wtns must stay valid (and unmodified) until its proof has been collected;
 * r32/s32 are copied at submit (NULL = random, drawn at collect).  On a sharded prover the pair is
 * zk_prove_dev_submit (r32/s32 ignored) + zk_prove_msm_collect, which hands back this shard's
 * partial sums for zk_prove_finish. */
#define ZK_MAX_IN_FLIGHT 8
int zk_prove_dev_submit(zk_prover *p, const void *d_wtns, const uint8_t *r32, const uint8_t *s32);
/* The same with the witness in HOST memory — the reference's own contract, Prover::prove(FrElement
 * *wtns) (src/groth16.hpp:101, call sites src/main_prover.cpp:74-75, src/fullprover.cpp:155).  The
 * upload runs on a stream of its own into a per-proof HBM buffer, so the witness of proof k+1 goes
 * up while proof k computes, and the call itself returns at once: a pageable buffer is copied to
 * pinned staging by a host function on that stream, a buffer from zk_host_alloc (or otherwise
 * page-locked) is read by the DMA engine directly.  Either way `wtns` must stay valid and untouched
 * until its proof has been collected (as for zk_prove_dev_submit). */
int zk_prove_submit(zk_prover *p, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32);
/* Page-locked host memory for witnesses (what a witness generator or a .wtns reader should fill:
 * src/fullprover.cpp:139-145 reads the file into a malloc'ed image, src/binfile_utils.cpp:28-33). */
int zk_host_alloc(void **out, size_t bytes);
void zk_host_free(void *ptr);
int zk_prove_collect(zk_prover *p, zk_proof *out);
/* Per-proof workspace is allocated the first time a slot / lane is used (the one-shot CLI never needs more than one).  A
 * server that will keep `in_flight` proofs in flight calls this once after create: every proof slot and lane such a
 * pipeline walks (in_flight + 1 slots of the ring, at most ZK_MAX_IN_FLIGHT; slot 0 alone for 1) is allocated NOW — with
 * host_witnesses != 0 including the per-slot HBM witness buffer and its pinned staging copy — so that running out of
 * device memory is a start-up error the caller can react to (tables as in the zkey instead of the window-precomputed
 * ones) and never a failed proof later.  Replaces nothing in the reference (its workspace is `new FrElement[]` per proof,
 * src/groth16.cpp:52-60). */
int zk_prover_reserve(zk_prover *p, uint32_t in_flight, uint32_t host_witnesses);
/* What zk_prover_create decided for this key on this device — the launch plan a benchmark, a server or a log line reports
 * (and sizes its pipeline by) instead of re-deriving the library's rules.  Set plan->size = sizeof(zk_prover_plan) before
 * the call (fields may be appended in later versions; only `size` bytes are written).  Replaces nothing in the
 * reference: its plan is fixed in code (one proof at a time, src/fullprover.cpp:96-97; window chosen inside ffiasm). */
typedef struct zk_prover_plan {
    uint32_t size;
    uint32_t window_bits_h, windows_h;     /* Pippenger window c of MSM H; digits = point additions per point */
    uint32_t window_bits_w, windows_w;     /* the same for the witness MSMs A, B1, B2, C (differs with ZK_FLAG_SPARSE_WITNESS) */
    uint32_t precomputed_tables;           /* 0: tables as in the zkey; 1: ZK_FLAG_PRECOMP (a row per window); 2: ZK_FLAG_PRECOMP_HALF (a row per second window) */
    uint32_t msm_a_b1_c_one_launch;        /* MSM A, B1, C as ONE set of launches over three tables (blockIdx.y) */
    uint32_t lanes;                        /* independent sets of compute streams: proof k runs on lane k % lanes */
    uint32_t follow_up_streams;            /* high-priority streams for merges / reductions (sharded provers) */
    uint32_t max_in_flight;                /* ZK_MAX_IN_FLIGHT */
    uint32_t depth_host_witness;           /* proofs in flight that saturate this prover: witnesses in host memory ... */
    uint32_t depth_resident_witness;       /* ... and already resident in HBM (no upload to hide) */
    uint32_t batch;                        /* opts.batch in effect (1 = single-witness submissions) */
    uint32_t shard_index, shard_count, chain_partitioned;
    uint64_t device_bytes_in_use;          /* HBM in use on the prover's device right now (all processes), from the runtime */
    uint64_t device_bytes_total;
    uint64_t kernel_launches_last_proof;   /* kernel launches the most recently submitted proof took (0 before the first; a graph replay counts as none) */
    uint32_t table_rows_h, table_rows_w;   /* rows of n points per table (x the zkey's table memory): windows_* with ZK_FLAG_PRECOMP, ceil(windows_* / 2) with _HALF, else 1 */
    uint32_t bucket_sets_h, bucket_sets_w; /* bucket sets (= bucket reductions) per MSM: 1 with ZK_FLAG_PRECOMP, 2 with _HALF, windows_* with plain tables */
} zk_prover_plan;
int zk_prover_info(zk_prover *p, zk_prover_plan *plan);
int zk_prove_msm_collect(zk_prover *p, zk_msm_sums *partial);
/* Batched proving (a prover created with opts.batch = B >= 2): `count` (1..B) witnesses of the circuit, given as
 * `count` host pointers, are proved by ONE submission — one digit sort with a bucket set per witness, one set of
 * accumulation / merge / reduction launches over all of them, the A.w/B.w rows and the transforms batched — which is
 * what a server for Semaphore-class circuits wants: there a proof is a chain of ~60 latency-bound kernels, and four
 * proofs in one chain cost little more than one.  The reference has no counterpart (one
 * Prover::prove per request, src/fullprover.cpp:154-159); every proof is the one zk_prove would give for the same
 * (witness, r, s).  r32s / s32s: count x 32 bytes or NULL (drawn at collect).  A submission occupies one of the
 * ZK_MAX_IN_FLIGHT slots; zk_prove_batch_collect takes the OLDEST submission, which must carry `count` proofs.  The
 * single-witness entry points work on a batch prover too (a submission of one). */
int zk_prove_batch_submit(zk_prover *p, const uint8_t *const *wtns, uint32_t count, const uint8_t *r32s, const uint8_t *s32s);
int zk_prove_batch_collect(zk_prover *p, zk_proof *out, uint32_t count);

/* Multi-GPU split of prove(): steps 1-10 (src/groth16.cpp:52-204) on this prover's shard ... */
int zk_prove_msm_dev(zk_prover *p, const void *d_wtns, zk_msm_sums *partial);
int zk_prove_msm(zk_prover *p, const uint8_t *wtns, zk_msm_sums *partial);
/* ... and steps 11-13 (src/groth16.cpp:209-253) over the partial sums of all shards (host, O(1)). */
int zk_prove_finish(zk_prover *p, const zk_msm_sums *partials, uint32_t n_partials,
                    const uint8_t *r32, const uint8_t *s32, zk_proof *out);

/* ---- one proof on several GPUs with the chain PARTITIONED too (north_star: "the five MSMs and the NTT
 * partitioned across the 8 GPUs").  GPU g holds rows [g*n/G, (g+1)*n/G) of a = A.w, b = B.w, c, h (the
 * slice its H table covers), runs the stages over the low log2(n/G) index bits of the six transforms
 * (src/groth16.cpp:98-155) locally and meets the others only in the log2(G) top stages: one radix-G
 * butterfly per block offset, for which each GPU receives 1/G of every block (all-to-all), computes,
 * and returns the results (all-to-all) — 2 x (G-1)/G of a block per GPU and transform over xGMI.
 *
 * (a) All GPUs in ONE process — what the reference's CLI / server are (src/main_prover.cpp:57-75,
 *     src/fullprover.cpp:154-159).  One shard prover per device inside; blocks exchanged by peer
 *     writes, cross-device ordering by events; one host thread enqueues everything.  devices may name
 *     the same device several times (all shards on one GPU: how the tests exercise this path on a
 *     1-GPU box).  The chain is partitioned when n_devices is 2, 4 or 8 and domainSize >= n_devices^2
 *     (ZKHIP_REPLICATED_CHAIN=1 in the environment keeps it replicated), otherwise replicated.
 *     zk_multi_prove = Prover::prove; submit/collect as zk_prove_submit / zk_prove_collect. */
typedef struct zk_multi_prover zk_multi_prover;
int zk_multi_prover_create(zk_multi_prover **out, const zk_zkey_view *zkey, const int32_t *devices, uint32_t n_devices,
                           const zk_opts *opts /* device, shard_* ignored */);
void zk_multi_prover_destroy(zk_multi_prover *mp);
int zk_multi_prove(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out);
int zk_multi_prove_submit(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32);
int zk_multi_prove_collect(zk_multi_prover *mp, zk_proof *out);
int zk_multi_prover_info(zk_multi_prover *mp, uint32_t *n_shards, uint32_t *chain_partitioned);
/* (b) One process per GPU (torch.distributed over RCCL): a prover created with shard_index/shard_count and
 *     ZK_FLAG_PARTITIONED_CHAIN is driven step by step, and the CALLER moves the blocks between the steps
 *     with FOUR all_to_all_single per proof on two buffers it owns and registers here (3 polynomials x
 *     block_elems x 32 bytes each, both laid out [GPU][polynomial][chunk]: `send` is what the library packs
 *     for / unpacks from the collective, `recv` is what the cross stages work on in place):
 *         zk_shard_begin                       rows of a, b, c; packed -> send     all_to_all_single(recv <- send)
 *         zk_shard_step(ZK_STEP_CROSS_INVERSE) top stages in place in recv         all_to_all_single(send <- recv)
 *         zk_shard_step(ZK_STEP_LOCAL)         unpack, local stages + coset, pack  all_to_all_single(recv <- send)
 *         zk_shard_step(ZK_STEP_CROSS_FORWARD) top stages in place in recv         all_to_all_single(send <- recv)
 *         zk_shard_step(ZK_STEP_FINISH)        unpack, h, MSM H, MSM C, joins      zk_prove_msm_collect + zk_prove_finish
 *     `stream` (hipStream_t; NULL = the default stream) is the stream the caller's collectives are ordered
 *     on: every call first waits for what is enqueued on it and makes it wait for what the call enqueued. */
enum { ZK_STEP_CROSS_INVERSE = 1, ZK_STEP_LOCAL = 2, ZK_STEP_CROSS_FORWARD = 3, ZK_STEP_FINISH = 4 };
int zk_shard_info(zk_prover *p, uint64_t *block_elems, uint32_t *chain_partitioned);
int zk_shard_set_exchange(zk_prover *p, void *d_send, void *d_recv);
int zk_shard_begin(zk_prover *p, const uint8_t *wtns, const void *d_wtns, const uint8_t *r32, const uint8_t *s32, void *stream);
int zk_shard_step(zk_prover *p, int step, void *stream);

/* Same as zk_prove_finish without a prover object: pure host code, usable on a rank that owns
 * no GPU (e.g. a coordinator).  vk points as in zk_zkey_view. */
int zk_assemble(const void *vk_alpha1, const void *vk_beta1, const void *vk_beta2, const void *vk_delta1,
                const void *vk_delta2, const zk_msm_sums *partials, uint32_t n_partials,
                const uint8_t *r32, const uint8_t *s32, zk_proof *out);

/* Device times of the last prove, ms (needs ZK_FLAG_TIMINGS), from hipEvents on the library's own
 * streams.  Two streams overlap (h chain | witness MSMs), so stage walls are not additive;
 * ZK_T_G1_L1_KERNEL is the mean duration of the four launches of the G1 level-1 accumulation kernel of the
 * last proof (MSM A, B1, C, H; events immediately before/after each launch, on its stream), ZK_T_G2_L1_KERNEL
 * the one launch of the G2 kernel, ZK_T_WTNS_H2D the witness upload of a host-witness proof. */
enum {
    ZK_T_SPMV = 0, ZK_T_NTT, ZK_T_DIGITS_SORT, ZK_T_MSM_H, ZK_T_JOIN_WAIT, ZK_T_MSM_REDUCE,
    ZK_T_TOTAL_DEVICE, ZK_T_G1_L1_KERNEL, ZK_T_G2_L1_KERNEL, ZK_T_WTNS_H2D, ZK_T_COUNT
};
int zk_prover_timings(zk_prover *p, double *ms, uint32_t n);

/* ---- operator-level entry points (host pointers; staged through the device) ------------- */
/* out[i] = a[i]*b[i]*R^-1 mod r  — E.fr.mul (src/groth16.cpp:91-95). */
int zk_fr_mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n);
/* same over Fq — E.f1.mul */
int zk_fq_mul_vec(uint8_t *out, const uint8_t *a, const uint8_t *b, uint64_t n);
/* a = A.w, b = B.w: the coefficient accumulation of src/groth16.cpp:62-85 as an operator.  coefs = zkey
 * section 4 INCLUDING its leading u32 count (packed 44-byte records, src/groth16.hpp:27-35); wtns standard
 * form; a, b (domainSize x 32 B each) come back in the reference's Montgomery form. */
int zk_fr_coef_accumulate(uint8_t *a, uint8_t *b, const void *coefs, uint64_t nCoefs, uint32_t domainSize,
                          const uint8_t *wtns, uint32_t nVars);
/* In-place natural-order NTT over Fr, Montgomery in/out: inverse=0 -> FFT::fft, 1 -> FFT::ifft
 * (incl. 1/n) (src/groth16.cpp:102,115).  n must be a power of two <= 2^28. */
int zk_fr_ntt(uint8_t *data, uint64_t n, int inverse);
/* The whole a/b/c pipeline of src/groth16.cpp:88-163 on host vectors a,b (Montgomery, n each):
 * h[i] = fromMontgomery(A(w2n^(2i+1))*B(..) - C(..)), standard form. */
int zk_fr_abc_to_h(uint8_t *h, const uint8_t *a, const uint8_t *b, uint64_t n);
/* out = sum scalars[i]*bases[i] — Curve::multiMulByScalar (src/groth16.cpp:173,197) with
 * scalarSize = 32; result as AFFINE Montgomery (all-zero = infinity). */
int zk_msm_g1(uint8_t out[64], const uint8_t *bases, const uint8_t *scalars, uint64_t n);
int zk_msm_g2(uint8_t out[128], const uint8_t *bases, const uint8_t *scalars, uint64_t n);

/* ---- synthetic tables & single-point helpers (benchmark inputs) ------ */
/* out[i] = P0 + i*Q, affine Montgomery, generated on the GPU (host output buffer). */
int zk_synth_chain_g1(uint8_t *out, uint64_t n, const uint8_t p0[64], const uint8_t q[64]);
int zk_synth_chain_g2(uint8_t *out, uint64_t n, const uint8_t p0[128], const uint8_t q[128]);
/* Batch fixed-base multiplication out[i] = scalars[i] * base (affine Montgomery out, scalars n x 32 B
 * LE standard form, 0 -> all-zero infinity encoding), on the GPU.  What a Groth16 setup does with the
 * evaluations A_i(tau), B_i(tau), ... (snarkjs zkey sections 5-9 as consumed at src/main_prover.cpp:67-72):
 * with it trapdoor-valid keys at benchmark sizes take seconds. */
int zk_fixed_base_g1(uint8_t *out, const uint8_t base[64], const uint8_t *scalars, uint64_t n);
int zk_fixed_base_g2(uint8_t *out, const uint8_t base[128], const uint8_t *scalars, uint64_t n);
/* out = k*P — Curve::mulByScalar (src/groth16.cpp:223) on the host; k: 32 B LE standard form. */
int zk_g1_mul(uint8_t out[64], const uint8_t p[64], const uint8_t k[32]);
int zk_g2_mul(uint8_t out[128], const uint8_t p[128], const uint8_t k[32]);

/* ---- output formatting (src/groth16.cpp:268-301, src/main_prover.cpp:77-93) - */
/* Compact JSON exactly as nlohmann's operator<< prints Proof::toJson(); returns needed length
 * (excluding NUL); writes at most cap bytes incl. NUL. */
size_t zk_proof_to_json(const zk_proof *proof, char *buf, size_t cap);
/* public.json: ["w1",...,"wN"], or null when nPublic == 0 (reference quirk Q7). */
size_t zk_public_to_json(const uint8_t *wtns, uint32_t nPublic, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
