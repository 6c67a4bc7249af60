#include "wtns_utils.hpp"

#include <cstring>
#include <stdexcept>

namespace WtnsUtils {

std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f) {
    std::unique_ptr<Header> h(new Header());
    f->startReadSection(1);
    h->n8 = f->readU32LE();
    if (h->n8 != 32) throw std::invalid_argument("wtns: only 256-bit fields are supported");
    memcpy(h->prime.data(), f->read(h->n8), 32);
    h->nVars = f->readU32LE();
    f->endReadSection();
    return h;
}

}   // namespace WtnsUtils
