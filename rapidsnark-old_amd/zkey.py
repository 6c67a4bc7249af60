"""ZKeyUtils::loadHeader mirror (reference src/zkey_utils.hpp:11-36, src/zkey_utils.cpp:17-52)."""


class ZkeyHeader:
    __slots__ = ("n8q", "qPrime", "n8r", "rPrime", "nVars", "nPublic", "domainSize", "nCoefs",
                 "vk_alpha1", "vk_beta1", "vk_beta2", "vk_gamma2", "vk_delta1", "vk_delta2")


def load_zkey_header(f) -> ZkeyHeader:
    h = ZkeyHeader()
    f.startReadSection(1)
    if f.readU32LE() != 1:
        raise ValueError("zkey file is not groth16")
    f.endReadSection()
    f.startReadSection(2)
    h.n8q = f.readU32LE()
    h.qPrime = int.from_bytes(f.read(h.n8q), "little")
    h.n8r = f.readU32LE()
    h.rPrime = int.from_bytes(f.read(h.n8r), "little")
    h.nVars = f.readU32LE()
    h.nPublic = f.readU32LE()
    h.domainSize = f.readU32LE()
    h.vk_alpha1 = bytes(f.read(h.n8q * 2))
    h.vk_beta1 = bytes(f.read(h.n8q * 2))
    h.vk_beta2 = bytes(f.read(h.n8q * 4))
    h.vk_gamma2 = bytes(f.read(h.n8q * 4))
    h.vk_delta1 = bytes(f.read(h.n8q * 2))
    h.vk_delta2 = bytes(f.read(h.n8q * 4))
    f.endReadSection()
    h.nCoefs = f.getSectionSize(4) // (12 + h.n8r)      # zkey_utils.cpp:49
    return h
