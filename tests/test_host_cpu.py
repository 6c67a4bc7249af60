"""CPU: host logic of the product — the binfile/zkey/wtns mirrors, the C-ABI library's export
list, and the host-only entry points (JSON formatting, point scalar-mul, zk_assemble).
No compute entry point is called: those need a GPU and must FAIL loudly without one."""
import ctypes
import os
import re

import pytest

from conftest import CIRCUITS, ROOT, golden_bytes, golden_json, golden_path
from oracle import bn254 as bn, groth16_ref as g


def test_library_exports_every_symbol_of_the_header(zk):
    hdr = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    declared = set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", hdr))
    from rapidsnark_old_amd import lib as L
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = ctypes.CDLL(zk.library_path())
    for name in declared:
        assert hasattr(lib, name), name


def test_compute_entry_points_fail_loudly_without_a_gpu(zk):
    try:
        n = zk.device_count()
    except zk.ZkHipError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(zk.ZkHipError):
        zk.fr_mul_vec(bytes(32), bytes(32))
    with pytest.raises(zk.ZkHipError):
        zk.msm_g1(bytes(64), bytes(32))
    with pytest.raises(zk.ZkHipError):
        zk.Prover(golden_path("multiplier2", "circuit.zkey"))
    with pytest.raises(zk.ZkHipError):
        zk.fr_ntt(bytes(64), inverse=False)
    with pytest.raises(zk.ZkHipError):
        zk.fixed_base_g1(bn.g1_to_bytes(bn.G1.gen), [1, 2, 3])
    with pytest.raises(zk.ZkHipError):
        zk.synth_chain_g1(4, bn.g1_to_bytes(bn.G1.gen), bn.g1_to_bytes(bn.G1.gen))


@pytest.mark.parametrize("name", CIRCUITS)
def test_host_json_matches_golden(zk, name):
    meta = golden_json(name, "meta.json")
    assert zk.proof_to_json(bytes.fromhex(meta["proof_bytes"])) == golden_bytes(name, "proof.json").decode()
    wt = g.read_wtns(golden_bytes(name, "witness.wtns"))
    vals = b"".join(bn.int_to_le32(v) for v in wt["witness"])
    f = zk.open_existing(golden_path(name, "circuit.zkey"), "zkey", 1)
    h = zk.load_zkey_header(f)
    assert zk.public_to_json(vals, h.nPublic) == golden_bytes(name, "public.json").decode()


def test_public_json_null_quirk(zk):
    assert zk.public_to_json(bn.int_to_le32(1), 0) == "null"          # main_prover.cpp:85-92 (Q7)
    assert zk.public_to_json(bn.int_to_le32(1) + bn.int_to_le32(0), 1) == '["0"]'


def test_host_point_mul(zk):
    g1, g2 = bn.g1_to_bytes(bn.G1.gen), bn.g2_to_bytes(bn.G2.gen)
    for k in (0, 1, 2, 12345678901234567890, bn.R_MOD - 1, bn.R_MOD):
        assert zk.g1_mul(g1, k) == bn.g1_to_bytes(bn.G1.mul(bn.G1.gen, k))
    assert zk.g2_mul(g2, 987654321) == bn.g2_to_bytes(bn.G2.mul(bn.G2.gen, 987654321))
    assert zk.g1_mul(bytes(64), 5) == bytes(64)


@pytest.mark.parametrize("name", CIRCUITS)
def test_assemble_from_golden_msm_sums(zk, name):
    """zk_assemble = src/groth16.cpp:209-253 on the host: five MSM results -> proof bytes."""
    meta = golden_json(name, "meta.json")
    f = zk.open_existing(golden_path(name, "circuit.zkey"), "zkey", 1)
    h = zk.load_zkey_header(f)
    vk = {k: getattr(h, k) for k in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2")}
    sums = bytes.fromhex(meta["pih"] + meta["pi_a"] + meta["pib1"] + meta["pi_b"] + meta["pi_c"])
    assert zk.assemble(vk, [sums], int(meta["r"]), int(meta["s"])).hex() == meta["proof_bytes"]
    a, b = zk.assemble(vk, [sums]), zk.assemble(vk, [sums])              # random r,s: 31 bytes each
    assert a != b


def test_binfile_mirror(zk):
    data = golden_bytes("r1cs_n8", "circuit.zkey")
    f = zk.open_existing(data, "zkey", 1)
    h = zk.load_zkey_header(f)
    ref = g.read_zkey(data)
    assert (h.nVars, h.nPublic, h.domainSize, h.nCoefs) == (ref.nVars, ref.nPublic, ref.domainSize, len(ref.coefs))
    assert h.qPrime == bn.Q_MOD and h.rPrime == bn.R_MOD and h.n8q == 32 and h.n8r == 32
    assert f.getSectionSize(5) == ref.nVars * 64 and f.getSectionSize(7) == ref.nVars * 128
    assert f.getSectionSize(8) == (ref.nVars - ref.nPublic - 1) * 64 and f.getSectionSize(9) == ref.domainSize * 64
    assert bytes(f.getSectionData(5)[:64]) == bn.g1_to_bytes(ref.A[0])
    w = zk.open_existing(golden_bytes("r1cs_n8", "witness.wtns"), "wtns", 2)
    wh = zk.load_wtns_header(w)
    assert wh.n8 == 32 and wh.prime == bn.R_MOD and wh.nVars == ref.nVars
    # error behaviour (same texts as binfile_utils.cpp:37-44,72-82; thrown by value — quirk Q1 fixed)
    with pytest.raises(ValueError, match="Invalid file type. It should be zkey"):
        zk.open_existing(golden_bytes("r1cs_n8", "witness.wtns"), "zkey", 1)
    bad = bytearray(data)
    bad[4] = 9
    with pytest.raises(ValueError, match="Invalid version"):
        zk.open_existing(bytes(bad), "zkey", 1)
    with pytest.raises(IndexError, match="Section does not exist"):
        f.getSectionData(77)
    f.startReadSection(1)
    with pytest.raises(IndexError, match="Already reading"):
        f.startReadSection(2)
    f.readU32LE()
    f.endReadSection()
    f.startReadSection(2)
    with pytest.raises(IndexError, match="Invalid section size"):
        f.endReadSection()


def test_zkey_not_groth16_is_rejected(zk):
    data = bytearray(golden_bytes("multiplier2", "circuit.zkey"))
    f = zk.open_existing(bytes(data), "zkey", 1)
    off = f.sections[1][0][0]
    data[off] = 2                                   # protocol id != 1
    with pytest.raises(ValueError, match="zkey file is not groth16"):
        zk.load_zkey_header(zk.open_existing(bytes(data), "zkey", 1))


def test_synthetic_family_definition(zk):
    import numpy as np
    from rapidsnark_old_amd import synth
    k = 8
    img = synth.make_coefs(k, 1, 0)
    n = 1 << k
    ncoefs = int(np.frombuffer(img[:4].tobytes(), dtype="<u4")[0])
    assert ncoefs == 4 * (n - 2) + 2 and img.size == 4 + 44 * ncoefs
    rec = np.frombuffer(img[4:].tobytes(), dtype=synth.COEF_DTYPE)
    assert set(np.unique(rec["m"])) == {0, 1} and rec["c"].max() == n - 1 and rec["s"].max() < n
    w = synth.make_witness(k)
    vals = [int.from_bytes(w[i * 32:(i + 1) * 32].tobytes(), "little") for i in range(n)]
    assert vals[0] == 1 and max(vals) < bn.R_MOD and len(set(vals)) > n - 3
    s0, s1 = synth.weighted_sums(w)
    assert s0 == sum(vals) and s1 == sum(i * v for i, v in enumerate(vals))
    assert synth.g1_gen_bytes() == bn.g1_to_bytes(bn.G1.gen) and synth.g2_gen_bytes() == bn.g2_to_bytes(bn.G2.gen)


def test_circuit_shaped_member_of_the_family(zk):
    """shape="circuit": nVars = 3/4 of the domain + 5, three public signals, ~30 % of the rows of A and of B1/B2 (the same
    rows) at infinity — and the known-discrete-log bookkeeping still predicts the C restatement's MSM sums."""
    import numpy as np
    from oracle import c_oracle as co
    from rapidsnark_old_amd import synth
    k = 9
    wl = synth.workload(k, co.chainp_g1, co.chainp_g2, co.g1_mul, co.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes(), shape="circuit")
    nv = wl["nVars"]
    assert nv == 3 * (1 << k) // 4 + 5 and wl["nPublic"] == 3 and wl["domainSize"] == 1 << k
    assert wl["pointsA"].size == nv * 64 and wl["pointsB2"].size == nv * 128 and wl["pointsC"].size == (nv - 4) * 64
    za, zb = wl["zero_rows"]["A"], wl["zero_rows"]["B"]
    assert 0.2 < za.mean() < 0.4 and 0.2 < zb.mean() < 0.4 and not (za == zb).all()
    A, B1, B2 = (np.asarray(wl[t]).reshape(nv, -1) for t in ("pointsA", "pointsB1", "pointsB2"))
    assert (A[za] == 0).all() and (A[~za].max(axis=1) > 0).all() and (B1[zb] == 0).all() and (B2[zb] == 0).all() and (B2[~zb].max(axis=1) > 0).all()
    rec = np.frombuffer(np.asarray(wl["coefs"])[4:].tobytes(), dtype=synth.COEF_DTYPE)
    assert rec["s"].max() < nv and rec.size == wl["nCoefs"]
    w = synth.make_witness(k, seed=2, kind="realistic", n_vars=nv)
    assert w.size == nv * 32
    sums = co.prove_msm(co.ZkeyView(wl), w)                  # zk_msm_sums: pih | pi_a | pib1 | pi_b (G2) | pi_c, affine points
    want = synth.expected_msm_dlogs(wl, w, np.zeros(32, dtype=np.uint8))
    G = bn.g1_to_bytes(bn.G1.gen)
    assert sums[64:128] == co.g1_mul(G, want["pi_a"]) and sums[128:192] == co.g1_mul(G, want["pib1"])
    assert sums[192:320] == co.g2_mul(bn.g2_to_bytes(bn.G2.gen), want["pi_b"]) and sums[320:384] == co.g1_mul(G, want["pi_c"])


def test_parity_kit_shim_serves_r_then_s(tmp_path):
    """tools/refcheck: the LD_PRELOAD replacement of randombytes_buf hands out r on the first call and s
    on the second, 31 bytes each, as src/groth16.cpp:216-217 consumes them."""
    import subprocess
    so = tmp_path / "librandshim.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-o", str(so), os.path.join(ROOT, "tools", "refcheck", "randombytes_shim.c")])
    r, s = 0x1122334455, (1 << 247) + 99
    code = ("import ctypes,os;l=ctypes.CDLL(%r);b=ctypes.create_string_buffer(31);"
            "l.randombytes_buf(b,ctypes.c_size_t(31));print(b.raw.hex());l.randombytes_buf(b,ctypes.c_size_t(31));print(b.raw.hex())" % str(so))
    env = dict(os.environ, ZKREF_R=r.to_bytes(32, "little").hex(), ZKREF_S=s.to_bytes(32, "little").hex())
    out = subprocess.run(["python3", "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == [r.to_bytes(31, "little").hex(), s.to_bytes(31, "little").hex()]


def test_parity_kit_files_mode_oracle_side():
    """tools/refcheck/refcheck.py --files ZKEY WTNS (snarkjs-made files against a real rapidsnark binary):
    the side the binary's output is compared with — the C restatement + the library's host-only JSON writers — reproduces the
    committed Multiplier2 fixture byte for byte when given the fixture's (r, s)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("refcheck_kit", os.path.join(ROOT, "tools", "refcheck", "refcheck.py"))
    kit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kit)
    d = os.path.join(ROOT, "tests", "golden", "multiplier2")
    meta = json.load(open(os.path.join(d, "meta.json")))
    pj, qj = kit.oracle_jsons(os.path.join(d, "circuit.zkey"), os.path.join(d, "witness.wtns"), int(meta["r"]), int(meta["s"]))
    assert pj == open(os.path.join(d, "proof.json"), "rb").read()
    assert qj == open(os.path.join(d, "public.json"), "rb").read()


def test_assemble_random_scalars_against_the_group_law(zk):
    """The tail's scalar multiplications use fixed-base tables for delta and one joint signed-window pass
    for s*pi_a + r*pi_b1 (csrc/host_tail.cpp): check the assembled proof against groth16.cpp:222-246
    evaluated with the oracle's plain double-and-add, for scalars that exercise the digit recoding
    (all-ones nibbles, carries into the 65th digit, zero) and for sums at infinity."""
    import random
    rnd = random.Random(7)
    G1, G2 = bn.G1, bn.G2
    pt1 = lambda: G1.mul(G1.gen, rnd.randrange(1, bn.R_MOD))
    pt2 = lambda: G2.mul(G2.gen, rnd.randrange(1, bn.R_MOD))
    alpha1, beta1, delta1, beta2, delta2 = pt1(), pt1(), pt1(), pt2(), pt2()
    vk = {"vk_alpha1": bn.g1_to_bytes(alpha1), "vk_beta1": bn.g1_to_bytes(beta1), "vk_beta2": bn.g2_to_bytes(beta2),
          "vk_delta1": bn.g1_to_bytes(delta1), "vk_delta2": bn.g2_to_bytes(delta2)}
    scalars = [(0, 0), (1, 0), (0, 1), (2**248 - 1, 2**248 - 1), (int("8" * 62, 16), int("7" * 62, 16)),
               (bn.R_MOD - 1, bn.R_MOD - 2), (2**256 - 1, 2**255 + 12345)]
    scalars += [(rnd.randrange(2**248), rnd.randrange(2**248)) for _ in range(6)]
    for i, (r, s) in enumerate(scalars):
        inf = i % 5 == 4
        h, a, b1, c = (None if inf else pt1() for _ in range(4))
        b2 = None if inf else pt2()
        sums = bn.g1_to_bytes(h) + bn.g1_to_bytes(a) + bn.g1_to_bytes(b1) + bn.g2_to_bytes(b2) + bn.g1_to_bytes(c)
        A = G1.add(G1.add(a, alpha1), G1.mul(delta1, r))
        B = G2.add(G2.add(b2, beta2), G2.mul(delta2, s))
        B1 = G1.add(G1.add(b1, beta1), G1.mul(delta1, s))
        C = G1.add(G1.add(c, h), G1.add(G1.mul(A, s), G1.mul(B1, r)))
        C = G1.sub(C, G1.mul(delta1, r * s % bn.R_MOD))
        want = bn.g1_to_bytes(A) + bn.g2_to_bytes(B) + bn.g1_to_bytes(C)
        assert zk.assemble(vk, [sums], r, s) == want, (i, r, s)


def test_tail_pool_runs_every_item_once_under_concurrent_callers(tmp_path):
    """csrc/tail_pool.hpp (the host tails of a batched submission run on it, several provers share it): six caller threads,
    thousands of for_each calls with 1..8 items — every item exactly once, no call returns early; a second build under
    ThreadSanitizer must stay silent (skipped where the sanitizer runtime is not installed)."""
    import subprocess
    src = os.path.join(ROOT, "tools", "tail_pool_test.cpp")
    inc = os.path.join(ROOT, "rapidsnark-old_amd", "csrc")
    exe = str(tmp_path / "tail_pool_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", inc, src, "-o", exe])
    out = subprocess.run([exe, "6", "3000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and " 0 wrong counts" in out.stdout, out.stdout + out.stderr
    tsan = str(tmp_path / "tail_pool_test_tsan")
    if subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", inc, src, "-o", tsan], capture_output=True).returncode == 0:
        out = subprocess.run([tsan, "6", "800"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr and " 0 wrong counts" in out.stdout, out.stdout + out.stderr[-2000:]


def test_column_bounds_of_the_29_bit_limb_products():
    """field29.hpp's Montgomery products accumulate a whole column in ONE signed 64-bit register.  With the lower eight reduction
    digits left as signed 32-bit values (only the top one is masked) a column holds, at worst, the operand terms the callers'
    limb bounds allow PLUS 2^31 x (sum of the modulus limbs): it must stay below 2^63 for both fields and every job shape."""
    src = open(os.path.join(ROOT, "rapidsnark-old_amd", "csrc", "field29.hpp")).read()
    for name, mod in (("Fq29Params", bn.Q_MOD), ("Fr29Params", bn.R_MOD)):
        limbs = [(mod >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
        decl = re.search(r"struct %s \{.*?P\[9\] = \{([^}]*)\}" % name, src, re.S).group(1)
        assert [int(x) for x in decl.split(",")] == limbs                     # the header's limbs ARE the modulus
        reduction = (1 << 31) * sum(limbs)
        tight, wide = (1 << 29) + 16, (1 << 30) + 32
        assert reduction + 18 * tight * tight + (1 << 35) < 1 << 63           # a*b + c*d (and the four-product job: <= 18 terms of one sign)
        assert reduction + 9 * tight * wide + (1 << 35) < 1 << 63             # a*b with ONE lazily added operand (DIT butterflies)
        assert reduction + (4 * wide * tight + tight * tight) + (1 << 35) < 1 << 63      # a^2 with the doubled operand


def test_no_product_column_leaves_int64_for_the_job_shapes_in_use(tmp_path):
    """Companion of the analytic bound above: tools/field29_bounds_test.cpp builds field29.hpp / curve29.hpp for the host with every
    multiply-accumulate evaluated in 128 bits (-DZK_CHECK_COLUMNS) and runs each job shape the kernels use (a*b, a*wide, a^2,
    a*b + c*d, the four-product job, chains of G1 / G2 mixed additions) over extreme and random operands of the callers' limb ranges."""
    import subprocess
    exe = str(tmp_path / "field29_bounds_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DZK_CHECK_COLUMNS", "-I", os.path.join(ROOT, "rapidsnark-old_amd", "csrc"),
                           os.path.join(ROOT, "tools", "field29_bounds_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK: no column left int64" in out.stdout, out.stdout + out.stderr
