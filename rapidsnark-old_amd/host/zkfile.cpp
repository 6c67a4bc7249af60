#include "zkfile.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <system_error>
#include <unistd.h>

namespace {

[[noreturn]] void sysfail(const char *what, int fd) {
    int e = errno;
    if (fd >= 0) close(fd);
    throw std::system_error(e, std::generic_category(), what);
}

template <class T>
T load_le(const uint8_t *p) {   // x86-64 / little-endian host
    T v;
    memcpy(&v, p, sizeof v);
    return v;
}

}   // namespace

namespace BinFileUtils {

BinFile::BinFile(const std::string &fileName, const std::string &type, uint32_t maxVersion) {
    const int fd = open(fileName.c_str(), O_RDONLY);
    if (fd < 0) sysfail("open", -1);
    struct stat st;
    if (fstat(fd, &st) != 0) sysfail("fstat", fd);
    mapLen_ = static_cast<uint64_t>(st.st_size);
    if (mapLen_) {
        void *m = mmap(nullptr, mapLen_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) sysfail("mmap", fd);
        map_ = static_cast<uint8_t *>(m);
    }
    close(fd);
    indexSections(type, maxVersion);
}

BinFile::BinFile(std::string &&image, const std::string &type, uint32_t maxVersion) : owned_(std::move(image)) {
    map_ = reinterpret_cast<uint8_t *>(owned_.data());
    mapLen_ = owned_.size();
    indexSections(type, maxVersion);
}

void BinFile::indexSections(const std::string &type, uint32_t maxVersion) {
    // preamble: 4-byte magic, u32 version, u32 section count
    const std::string magic(reinterpret_cast<const char *>(take(4)), 4);
    if (magic != type) throw std::invalid_argument("Invalid file type. It should be " + type + " and it us " + magic);
    const uint32_t version = readU32LE();
    if (version > maxVersion)
        throw std::invalid_argument("Invalid version. It should be <=" + std::to_string(maxVersion) + " and it us " + std::to_string(version));
    // directory: { u32 type, u64 length, payload } repeated; duplicates of a type are kept in order
    for (uint32_t left = readU32LE(); left > 0; left--) {
        const uint32_t id = readU32LE();
        const uint64_t len = readU64LE();
        const uint64_t at = cursor_;
        take(len);
        index_[id].push_back(Extent{at, len});
    }
    cursor_ = 0;
}

BinFile::~BinFile() {
    if (map_ && owned_.empty()) munmap(map_, mapLen_);
}

const uint8_t *BinFile::take(uint64_t len) {
    if (len > mapLen_ || cursor_ > mapLen_ - len) throw std::range_error("Unexpected end of file");
    const uint8_t *p = map_ + cursor_;
    cursor_ += len;
    return p;
}

const BinFile::Extent &BinFile::extent(uint32_t id, uint32_t nth) const {
    const auto hit = index_.find(id);
    if (hit == index_.end()) throw std::range_error("Section does not exist: " + std::to_string(id));
    const auto &list = hit->second;
    if (nth >= list.size())
        throw std::range_error("Section pos too big. There are " + std::to_string(list.size()) +
                               " and it's trying to access section: " + std::to_string(nth));
    return list[nth];
}

void BinFile::startReadSection(uint32_t sectionId, uint32_t sectionPos) {
    const Extent &e = extent(sectionId, sectionPos);
    if (open_) throw std::range_error("Already reading a section");
    open_ = &e;
    cursor_ = e.begin;
}

void BinFile::endReadSection(bool check) {
    const Extent *e = open_;
    open_ = nullptr;
    if (check && e && cursor_ != e->begin + e->length) throw std::range_error("Invalid section size");
}

uint32_t BinFile::readU32LE() { return load_le<uint32_t>(take(4)); }
uint64_t BinFile::readU64LE() { return load_le<uint64_t>(take(8)); }
void *BinFile::read(uint64_t len) { return const_cast<uint8_t *>(take(len)); }

void *BinFile::getSectionData(uint32_t sectionId, uint32_t sectionPos) { return map_ + extent(sectionId, sectionPos).begin; }
uint64_t BinFile::getSectionSize(uint32_t sectionId, uint32_t sectionPos) { return extent(sectionId, sectionPos).length; }

std::unique_ptr<BinFile> openExisting(const std::string &filename, const std::string &type, uint32_t maxVersion) {
    return std::make_unique<BinFile>(filename, type, maxVersion);
}
std::unique_ptr<BinFile> fromMemory(std::string &&image, const std::string &type, uint32_t maxVersion) {
    return std::make_unique<BinFile>(std::move(image), type, maxVersion);
}

}   // namespace BinFileUtils

namespace {

// u32 byte-length followed by that many little-endian bytes; only 256-bit fields exist on this path
uint32_t read_field_modulus(BinFileUtils::BinFile &f, std::array<uint8_t, 32> &out, const char *who) {
    const uint32_t n8 = f.readU32LE();
    if (n8 != out.size()) throw std::invalid_argument(std::string(who) + ": only 256-bit fields are supported");
    memcpy(out.data(), f.read(n8), n8);
    return n8;
}

}   // namespace

namespace ZKeyUtils {

std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f) {
    auto h = std::make_unique<Header>();

    f->startReadSection(1);   // protocol id: 1 = groth16
    const bool groth16 = f->readU32LE() == 1;
    if (!groth16) throw std::invalid_argument("zkey file is not groth16");
    f->endReadSection();

    f->startReadSection(2);
    h->n8q = read_field_modulus(*f, h->qPrime, "zkey");
    h->n8r = read_field_modulus(*f, h->rPrime, "zkey");
    for (uint32_t *dst : {&h->nVars, &h->nPublic, &h->domainSize}) *dst = f->readU32LE();
    // verification-key points, in file order; G1 = 2 coordinates, G2 = 4
    struct { void **slot; uint32_t coords; } const vk[] = {{&h->vk_alpha1, 2}, {&h->vk_beta1, 2}, {&h->vk_beta2, 4},
                                                           {&h->vk_gamma2, 4}, {&h->vk_delta1, 2}, {&h->vk_delta2, 4}};
    for (const auto &p : vk) *p.slot = f->read(uint64_t(h->n8q) * p.coords);
    f->endReadSection();

    // section 4 = u32 count + packed {u32 m, u32 c, u32 s, Fr coef}; the count's 4 bytes vanish in the division
    h->nCoefs = f->getSectionSize(4) / (3 * sizeof(uint32_t) + h->n8r);
    return h;
}

}   // namespace ZKeyUtils

namespace WtnsUtils {

std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f) {
    auto h = std::make_unique<Header>();
    f->startReadSection(1);
    h->n8 = read_field_modulus(*f, h->prime, "wtns");
    h->nVars = f->readU32LE();
    f->endReadSection();
    return h;
}

}   // namespace WtnsUtils
