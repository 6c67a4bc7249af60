"""WtnsUtils::loadHeader mirror (reference src/wtns_utils.hpp:10-21, src/wtns_utils.cpp:12-25)."""


class WtnsHeader:
    __slots__ = ("n8", "prime", "nVars")


def load_wtns_header(f) -> WtnsHeader:
    h = WtnsHeader()
    f.startReadSection(1)
    h.n8 = f.readU32LE()
    h.prime = int.from_bytes(f.read(h.n8), "little")
    h.nVars = f.readU32LE()
    f.endReadSection()
    return h
