#include "zkey_utils.hpp"

#include <cstring>
#include <stdexcept>

namespace ZKeyUtils {

static void read_prime(BinFileUtils::BinFile *f, uint32_t n8, std::array<uint8_t, 32> &out) {
    if (n8 != 32) throw std::invalid_argument("zkey: only 256-bit fields are supported");
    memcpy(out.data(), f->read(n8), 32);
}

std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f) {
    std::unique_ptr<Header> h(new Header());

    f->startReadSection(1);
    uint32_t protocol = f->readU32LE();
    if (protocol != 1) throw std::invalid_argument("zkey file is not groth16");
    f->endReadSection();

    f->startReadSection(2);
    h->n8q = f->readU32LE();
    read_prime(f, h->n8q, h->qPrime);
    h->n8r = f->readU32LE();
    read_prime(f, h->n8r, h->rPrime);
    h->nVars = f->readU32LE();
    h->nPublic = f->readU32LE();
    h->domainSize = f->readU32LE();
    h->vk_alpha1 = f->read(h->n8q * 2);
    h->vk_beta1 = f->read(h->n8q * 2);
    h->vk_beta2 = f->read(h->n8q * 4);
    h->vk_gamma2 = f->read(h->n8q * 4);
    h->vk_delta1 = f->read(h->n8q * 2);
    h->vk_delta2 = f->read(h->n8q * 4);
    f->endReadSection();

    // 44-byte packed records after a u32 count; the 4 leading bytes vanish in the division
    h->nCoefs = f->getSectionSize(4) / (12 + h->n8r);
    return h;
}

}   // namespace ZKeyUtils
