"""zkgen — trapdoor-VALID Groth16 keys at benchmark sizes, product side.

Writes a real snarkjs-format `circuit.zkey` (sections as consumed by the reference,
src/zkey_utils.cpp:17-52, src/main_prover.cpp:57-72), a satisfying `witness.wtns`
(src/wtns_utils.cpp:12-25) and a snarkjs `verification_key.json`, for a random R1CS with a 2^k
domain — so that the BENCHMARK configurations can be proved on valid keys and the proofs verified
off-box with `snarkjs groth16 verify verification_key.json public.json proof.json`.

Everything heavy runs on the GPU through the product's operator-level C-ABI; this module holds no
field or curve arithmetic of its own beyond Python integers for a handful of scalars:
  * the Lagrange basis at tau is ONE inverse NTT of the powers of tau
    (L_j(tau) = 1/n sum_k tau^k w^-jk)                                              zk_fr_ntt
  * A_i(tau), B_i(tau), K_i = beta A_i + alpha B_i + C_i are the coefficient
    accumulation of src/groth16.cpp:62-85 run on the TRANSPOSED records           zk_fr_coef_accumulate
  * the witness's internal signals are (A.w) o (B.w)                              zk_fr_coef_accumulate, zk_fr_mul_vec
  * the five point tables and IC are batch fixed-base multiplications            zk_fixed_base_g1/g2
The circuit: m constraints  (a1 w_p + a2 w_q) * (b1 w_u + b2) = w_out  over random input signals
(the first nPublic of them public), every constraint defining one internal signal; nVars = domainSize.

The toxic waste is returned (and written by the CLI): with it the discrete logs of a proof are
computable in Fr alone (`expected_proof_dlogs`), the pairing-free check of SURVEY §8(c) item 2.
Nothing here imports oracle/."""
import json
import os
import struct

import numpy as np

from . import lib as L
from . import synth

R_MOD = synth.R_MOD
Q_MOD = synth.Q_MOD
MONT = 1 << 256


def _le(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8)


def _const(x, n):
    return np.tile(_le(x % R_MOD), n)


def _mul(a, b):
    """a*b/R elementwise on the GPU (numpy uint8 in/out)."""
    return np.frombuffer(L.fr_mul_vec(a, b), dtype=np.uint8)


def _scale(a, k, n):
    """standard a[i]*k -> standard: (a*k/R) * R^2 / R"""
    return _mul(_mul(a, _const(k, n)), _const(MONT * MONT, n))


def _powers(tau, n):
    """tau^0 .. tau^(n-1), standard form, by doubling on the GPU."""
    t = _le(1).copy()
    length = 1
    while length < n:
        t = np.concatenate([t, _scale(t, pow(tau, length, R_MOD), length)])
        length *= 2
    return t[:n * 32]


def _records(m_col, c_col, s_col, vals):
    rec = np.zeros(len(c_col), dtype=synth.COEF_DTYPE)
    rec["m"], rec["c"], rec["s"] = m_col, c_col, s_col
    rec["v"] = vals.reshape(-1, 32)
    return rec


def _image(rec):
    img = np.empty(4 + rec.size * 44, dtype=np.uint8)
    img[:4] = np.frombuffer(np.uint32(rec.size).tobytes(), dtype=np.uint8)
    img[4:] = rec.view(np.uint8).reshape(-1)
    return img


def _sum_mod(vals):
    return synth.weighted_sums(vals)[0] % R_MOD


def _small_fr(rng, count, bits):
    """count values below 2^bits (bits <= 32) as standard-form 32-byte rows"""
    out = np.zeros((count, 32), dtype=np.uint8)
    out[:, :4] = rng.integers(0, 1 << bits, size=count, dtype=np.uint64).astype("<u4").view(np.uint8).reshape(count, 4)
    return out


def _semaphore_like_circuit(m, n_in, n_public, rng, prng):
    """Rows of a Semaphore-shaped R1CS in this generator's constraint form (a1 w_p + a2 w_q)(b1 w_u + b2) = w_out: a long chain
    of hash "levels" as in a Merkle-path circuit — per level one boolean path bit (b*b), a two-constraint mux of the running hash
    and a sibling (d = sib - cur, sel = b*d), one linear constraint forming the hash input, and ROUNDS rounds of an x^5 S-box with
    a round constant (t1 = (x+k)^2, t2 = t1^2, y = t2*(x+k)), each round reading the previous one's output.  Published shape of
    the Semaphore / iden3-auth class: almost every signal is a full-size field element, a few per cent are boolean selectors,
    constraints depend on each other in chains thousands deep, every row has one or two non-zero coefficients per matrix.
    -> (p, q, u, a1, a2, b1, b2) as in generate(), and the witness of the internal signals computed on the host in row order."""
    ROUNDS = 30
    one = 0                                        # signal 0 is the constant 1
    P, Q, U, A1, A2, B1, B2 = ([0] * m for _ in range(7))
    bits = list(range(1 + n_public, 1 + n_public + (n_in - n_public) // 2))            # boolean private inputs (path indices)
    sibs = list(range(1 + n_public + (n_in - n_public) // 2, 1 + n_in))                 # full-size private inputs (siblings, secrets)
    w_in = [0] * (1 + n_in)
    w_in[0] = 1
    for i in range(1, 1 + n_public):
        w_in[i] = prng.randrange(R_MOD)
    for i in bits:
        w_in[i] = prng.randrange(2)
    for i in sibs:
        w_in[i] = prng.randrange(R_MOD)
    val = list(w_in) + [0] * m                     # signal values; internal signal of row r is 1 + n_in + r

    def row(r, a1, p, a2, q, b1, u, b2):
        P[r], Q[r], U[r], A1[r], A2[r], B1[r], B2[r] = p, q, u, a1 % R_MOD, a2 % R_MOD, b1 % R_MOD, b2 % R_MOD
        val[1 + n_in + r] = (a1 * val[p] + a2 * val[q]) % R_MOD * ((b1 * val[u] + b2) % R_MOD) % R_MOD
        return 1 + n_in + r

    r, level, cur = 0, 0, sibs[0]
    while r < m:
        b, sib = bits[level % len(bits)], sibs[(level + 1) % len(sibs)]
        level += 1
        if r + 4 >= m:                             # not enough rows left for a level: pad with boolean rows
            row(r, 1, b, 0, one, 1, b, 0)
            r += 1
            continue
        row(r, 1, b, 0, one, 1, b, 0)                                                    # b*b (= b: the boolean check's product)
        d = row(r + 1, 1, sib, -1, cur, 0, one, 1)                                       # d = sib - cur          (linear)
        sel = row(r + 2, 1, b, 0, one, 1, d, 0)                                          # sel = b * d
        x = row(r + 3, 1, cur, 1, sel, 0, one, 1)                                        # x = cur + sel          (linear): the level's hash input
        r += 4
        for _ in range(ROUNDS):
            if r + 3 > m:
                break
            kc = prng.randrange(R_MOD)
            t1 = row(r, 1, x, kc, one, 1, x, kc)                                         # (x + k)^2
            t2 = row(r + 1, 1, t1, 0, one, 1, t1, 0)                                     # (x + k)^4
            x = row(r + 2, 1, t2, 0, one, 1, x, kc)                                      # (x + k)^5
            r += 3
        cur = x
    to_rows = lambda xs: np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in xs), dtype=np.uint8).copy()
    w_rows = to_rows(val[1:]).reshape(-1, 32)
    return (np.array(P, np.uint32), np.array(Q, np.uint32), np.array(U, np.uint32), to_rows(A1), to_rows(A2), to_rows(B1), to_rows(B2),
            w_rows[:n_in], w_rows[n_in:])


def generate(k, n_public=2, seed=0, circuit_like=False, semaphore_like=False):
    """-> dict with the zkey sections (numpy uint8), the witness, the trapdoor and the vectors the
    trapdoor check needs.  Needs a GPU.

    circuit_like: what circom circuits look like instead of uniformly random everything — nVars = 3/4 of the domain + 5
    (never the domain size); 80 % of the signals boolean (inputs drawn from {0,1}, internal signals the AND of two boolean
    inputs), 15 % small (16-bit coefficients on small inputs), 5 % full-size; a second layer of constraints reads the first
    layer's outputs; wires that never occur in A (or in B) leave all-zero rows = points at infinity in those tables.

    semaphore_like: the shape class of BASELINE configs[4] (Semaphore / iden3 auth; no such key exists in this image) — see
    _semaphore_like_circuit: 64 inputs, nVars = 3/4 of the domain + 5, chains of x^5 S-box rounds between Merkle-style muxes,
    nearly every signal a full-size field element; the witness is computed on the host (the constraints form one long chain)."""
    n = 1 << k
    n_in = 64 if semaphore_like else max(n_public + 1, n // 8)                 # input signals (incl. the public ones)
    if semaphore_like and n_in < n_public + 8:
        raise ValueError("semaphore_like needs at most 56 public signals")
    m = (3 * n // 4 + 5 if (circuit_like or semaphore_like) else n) - 1 - n_in      # constraints = internal signals
    if m < 1 or m + n_public + 1 > n:
        raise ValueError("domain too small")
    n_vars = 1 + n_in + m                            # = n (circuit_like: 3n/4 + 5)
    rng = np.random.default_rng(0x2C6E0000 + 131 * k + seed)
    import random
    prng = random.Random(0x70C51C + 977 * k + seed)
    tau, alpha, beta, gamma, delta = (prng.randrange(2, R_MOD) for _ in range(5))

    # ---- circuit
    p, q, u = (rng.integers(0, 1 + n_in, size=m, dtype=np.uint32) for _ in range(3))
    a1, a2, b1, b2 = (synth.random_fr_bytes(rng, m).reshape(-1) for _ in range(4))          # standard values
    w_in = synth.random_fr_bytes(rng, n_in)
    layer2 = np.zeros(m, dtype=bool)
    w_internal = None
    if semaphore_like:
        p, q, u, a1, a2, b1, b2, w_in, w_internal = _semaphore_like_circuit(m, n_in, n_public, rng, prng)
    if circuit_like:
        cls_in = rng.random(n_in)
        w_in[cls_in < 0.80] = _small_fr(rng, int((cls_in < 0.80).sum()), 1)
        mid_in = (cls_in >= 0.80) & (cls_in < 0.95)
        w_in[mid_in] = _small_fr(rng, int(mid_in.sum()), 32)
        bool_in = (1 + np.nonzero(cls_in < 0.80)[0]).astype(np.uint32)                      # signal numbers of the boolean inputs
        small_in = (1 + np.nonzero(cls_in < 0.95)[0]).astype(np.uint32)
        cls = rng.random(m)
        is_and, is_small = cls < 0.80, (cls >= 0.80) & (cls < 0.95)
        one, zero = np.zeros((1, 32), np.uint8), np.zeros((1, 32), np.uint8)
        one[0, 0] = 1
        A1, A2, B1c, B2c = (x.reshape(m, 32) for x in (a1, a2, b1, b2))
        # AND of two boolean inputs: (1 w_p + 0) * (1 w_u + 0) = w_out
        p[is_and] = rng.choice(bool_in, size=int(is_and.sum()))
        u[is_and] = rng.choice(bool_in, size=int(is_and.sum()))
        A1[is_and], A2[is_and], B1c[is_and], B2c[is_and] = one, zero, one, zero
        # small arithmetic: 16-bit coefficients on small inputs (products stay far below 2^128)
        ns = int(is_small.sum())
        p[is_small], q[is_small], u[is_small] = (rng.choice(small_in, size=ns) for _ in range(3))
        A1[is_small], A2[is_small], B1c[is_small], B2c[is_small] = (_small_fr(rng, ns, 16) for _ in range(4))
        # the rest: full-size coefficients; in the second half of the constraints they read the FIRST half's outputs
        layer2 = (~is_and) & (~is_small) & (np.arange(m) >= m // 2)
        nl2 = int(layer2.sum())
        p[layer2], q[layer2], u[layer2] = (rng.integers(1 + n_in, 1 + n_in + m // 2, size=nl2, dtype=np.uint32) for _ in range(3))
        a1, a2, b1, b2 = (x.reshape(-1) for x in (A1, A2, B1c, B2c))
    r3 = _const(MONT ** 3, m)
    a1m, a2m, b1m, b2m = (_mul(x, r3) for x in (a1, a2, b1, b2))                             # value * R^2: the zkey's coefficient form
    rows = np.arange(m, dtype=np.uint32)
    out_idx = (1 + n_in + rows).astype(np.uint32)
    one_r2 = _le(MONT * MONT % R_MOD)
    extra_rows = (m + np.arange(n_public + 1)).astype(np.uint32)
    recA = np.concatenate([_records(0, rows, p, a1m), _records(0, rows, q, a2m),
                           _records(0, extra_rows, np.arange(n_public + 1, dtype=np.uint32), np.tile(one_r2, n_public + 1))])
    recB = np.concatenate([_records(1, rows, u, b1m), _records(1, rows, np.zeros(m, np.uint32), b2m)])
    if circuit_like or semaphore_like:              # a real zkey holds no zero coefficients
        recA = recA[recA["v"].max(axis=1) > 0]
        recB = recB[recB["v"].max(axis=1) > 0]
    rec = np.concatenate([recA, recB])
    rec = rec[rng.permutation(rec.size)]            # the loader must not rely on any order
    coefs = _image(rec)

    # ---- witness: inputs random, internal signals = (A.w) o (B.w) (a second pass once the first layer's outputs exist)
    w = np.zeros((n_vars, 32), dtype=np.uint8)
    w[0, 0] = 1
    w[1:1 + n_in] = w_in
    for _pass in range(0 if w_internal is not None else (2 if layer2.any() else 1)):
        am, bm = L.fr_coef_accumulate(coefs, rec.size, n, w.reshape(-1))
        prod = _mul(_mul(am[:m * 32], bm[:m * 32]), _const(1, m))                            # (aR)(bR)/R /R = ab
        w[1 + n_in:] = prod.reshape(-1, 32)
    if w_internal is not None:
        w[1 + n_in:] = w_internal                    # the chain was evaluated row by row on the host
    w = w.reshape(-1)

    # ---- Fr half of the setup
    lag = np.frombuffer(L.fr_ntt(_powers(tau, n), inverse=True), dtype=np.uint8)             # L_j(tau), standard
    lag2 = np.frombuffer(L.fr_ntt(_powers(tau, 2 * n), inverse=True), dtype=np.uint8).reshape(-1, 32)
    dinv, ginv = pow(delta, -1, R_MOD), pow(gamma, -1, R_MOD)
    hs = _scale(np.ascontiguousarray(lag2[1::2]).reshape(-1), dinv, n)                       # L^(2n)_{2i+1}(tau) / delta
    del lag2
    # transposed records: "row" = signal, "signal" = constraint; the witness slot holds L(tau)
    tA, tB = recA.copy(), recB.copy()
    tA["c"], tA["s"] = recA["s"], recA["c"]
    tB["c"], tB["s"] = recB["s"], recB["c"]
    at_m, bt_m = L.fr_coef_accumulate(_image(np.concatenate([tA, tB])), tA.size + tB.size, n_vars, lag)
    at = _mul(at_m, _const(1, n_vars))                                                        # A_i(tau), standard
    bt = _mul(bt_m, _const(1, n_vars))
    # K_i = beta A_i + alpha B_i + C_i, C_i(tau) = L_j(tau) for the signal constraint j defines
    kA, kB = tA.copy(), tB.copy()
    kA["v"] = _mul(tA["v"].reshape(-1), _const(beta * MONT, tA.size)).reshape(-1, 32)
    kB["v"] = _mul(tB["v"].reshape(-1), _const(alpha * MONT, tB.size)).reshape(-1, 32)
    kB["m"] = 0
    kC = _records(0, out_idx, rows, np.tile(one_r2, m))
    k_m, _ = L.fr_coef_accumulate(_image(np.concatenate([kA, kB, kC])), kA.size + kB.size + kC.size, n_vars, lag)
    kk = _mul(k_m, _const(1, n_vars))                                                         # K_i, standard (trapdoor check)
    c_sc = _mul(k_m, _const(dinv, n_vars)).reshape(-1, 32)[n_public + 1:].reshape(-1)         # K_i / delta
    ic_sc = _mul(k_m, _const(ginv, n_vars)).reshape(-1, 32)[:n_public + 1].reshape(-1)        # K_i / gamma
    ct = np.zeros((n_vars, 32), dtype=np.uint8)
    ct[1 + n_in:] = lag.reshape(-1, 32)[:m]

    # ---- curve half: batch fixed-base multiplications
    g1, g2 = synth.g1_gen_bytes(), synth.g2_gen_bytes()
    u8 = lambda b: np.frombuffer(bytes(b), dtype=np.uint8)
    out = {
        "k": k, "nVars": n_vars, "nPublic": n_public, "domainSize": n, "nCoefs": int(rec.size), "coefs": coefs,
        "pointsA": L.fixed_base_g1(g1, at), "pointsB1": L.fixed_base_g1(g1, bt), "pointsB2": L.fixed_base_g2(g2, bt),
        "pointsC": L.fixed_base_g1(g1, c_sc), "pointsH": L.fixed_base_g1(g1, hs), "pointsIC": L.fixed_base_g1(g1, ic_sc),
        "vk_alpha1": u8(L.g1_mul(g1, alpha)), "vk_beta1": u8(L.g1_mul(g1, beta)), "vk_beta2": u8(L.g2_mul(g2, beta)),
        "vk_gamma2": u8(L.g2_mul(g2, gamma)), "vk_delta1": u8(L.g1_mul(g1, delta)), "vk_delta2": u8(L.g2_mul(g2, delta)),
        "witness": w,
        "trap": {"toxic": (tau, alpha, beta, gamma, delta), "At": at, "Bt": bt, "Ct": ct.reshape(-1), "K": kk},
    }
    return out


def expected_proof_dlogs(key, r, s):
    """Discrete logs (a, b, c) of the proof points pi_a, pi_b, pi_c for the key's own witness
    (SURVEY §8c item 2): a = alpha + sum w_i A_i(tau) + r delta, b = beta + sum w_i B_i(tau) + s delta,
    c = (sum_{i > nPublic} w_i K_i + H(tau) Z(tau)) / delta + s a + r b - r s delta."""
    t = key["trap"]
    tau, alpha, beta, gamma, delta = t["toxic"]
    w, npub, nv = key["witness"], key["nPublic"], key["nVars"]
    def dot_exact(vec, lo=0):          # sum vec_i * w_i over i >= lo: (v w / R) * R^2 / R = v w, summed exactly, then mod r
        prod = _mul(_mul(vec, w), _const(MONT * MONT, nv))
        return _sum_mod(prod.reshape(-1, 32)[lo:])

    da, db, dc = dot_exact(t["At"]), dot_exact(t["Bt"]), dot_exact(t["Ct"])
    a = (alpha + da + r * delta) % R_MOD
    b = (beta + db + s * delta) % R_MOD
    hz = (da * db - dc) % R_MOD
    dinv = pow(delta, -1, R_MOD)
    c = ((dot_exact(t["K"], npub + 1) + hz) * dinv + s * a + r * b - (r * s % R_MOD) * delta) % R_MOD
    return a, b, c


# ---------------------------------------------------------------- files
def _binfile(magic, version, sections):
    out = [magic, struct.pack("<II", version, len(sections))]
    for sid, payload in sections:
        payload = payload if isinstance(payload, (bytes, bytearray)) else np.ascontiguousarray(payload).tobytes()
        out.append(struct.pack("<IQ", sid, len(payload)))
        out.append(payload)
    return out


def write_zkey(key, path):
    b = lambda name: np.ascontiguousarray(key[name]).tobytes()
    sec2 = (struct.pack("<I", 32) + int(Q_MOD).to_bytes(32, "little") + struct.pack("<I", 32) + int(R_MOD).to_bytes(32, "little")
            + struct.pack("<III", key["nVars"], key["nPublic"], key["domainSize"])
            + b("vk_alpha1") + b("vk_beta1") + b("vk_beta2") + b("vk_gamma2") + b("vk_delta1") + b("vk_delta2"))
    parts = _binfile(b"zkey", 1, [(1, struct.pack("<I", 1)), (2, sec2), (3, key["pointsIC"]), (4, key["coefs"]),
                                  (5, key["pointsA"]), (6, key["pointsB1"]), (7, key["pointsB2"]), (8, key["pointsC"]),
                                  (9, key["pointsH"]), (10, bytes(68))])
    with open(path, "wb") as f:
        for p in parts:
            f.write(p)


def write_wtns(key, path):
    sec1 = struct.pack("<I", 32) + int(R_MOD).to_bytes(32, "little") + struct.pack("<I", key["nVars"])
    with open(path, "wb") as f:
        for p in _binfile(b"wtns", 2, [(1, sec1), (2, key["witness"])]):
            f.write(p)


def _fq_std(mont_bytes):
    """Montgomery Fq coordinates -> list of standard-form ints."""
    a = np.ascontiguousarray(mont_bytes).reshape(-1)
    n = a.size // 32
    one = np.tile(np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8), n)
    std = L.fq_mul_vec(a, one)
    return [int.from_bytes(std[32 * i:32 * i + 32], "little") for i in range(n)]


def verification_key(key):
    """snarkjs verification_key.json content (groth16_verify.js reads vk_alpha_1, vk_beta_2, vk_gamma_2,
    vk_delta_2 and IC; vk_alphabeta_12 — a pairing value it does not use for verification — is omitted:
    there is no pairing in this repository)."""
    g1 = lambda b: [str(v) for v in _fq_std(b)] + ["1"]

    def g2(b):
        xa, xb, ya, yb = _fq_std(b)
        return [[str(xa), str(xb)], [str(ya), str(yb)], ["1", "0"]]

    ic = np.ascontiguousarray(key["pointsIC"]).reshape(-1, 64)
    return {"protocol": "groth16", "curve": "bn128", "nPublic": key["nPublic"],
            "vk_alpha_1": g1(key["vk_alpha1"]), "vk_beta_2": g2(key["vk_beta2"]), "vk_gamma_2": g2(key["vk_gamma2"]),
            "vk_delta_2": g2(key["vk_delta2"]), "IC": [g1(row) for row in ic]}


def write_all(key, outdir):
    os.makedirs(outdir, exist_ok=True)
    write_zkey(key, os.path.join(outdir, "circuit.zkey"))
    write_wtns(key, os.path.join(outdir, "witness.wtns"))
    with open(os.path.join(outdir, "verification_key.json"), "w") as f:
        json.dump(verification_key(key), f, indent=1)
    with open(os.path.join(outdir, "toxic.json"), "w") as f:
        json.dump({name: str(v) for name, v in zip(("tau", "alpha", "beta", "gamma", "delta"), key["trap"]["toxic"])}, f)
