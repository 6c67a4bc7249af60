// zkfile — snarkjs binary containers (.zkey / .wtns) for the MI355X prover host.
//
// One translation unit provides the three namespaces the reference spreads over
// binfile_utils / zkey_utils / wtns_utils (src/binfile_utils.hpp:10-52, src/zkey_utils.hpp:11-36,
// src/wtns_utils.hpp:10-21) with the same public names, so code written against the reference
// (`BinFileUtils::openExisting`, `BinFile::getSectionData`, `ZKeyUtils::loadHeader`, ...) compiles
// against it unchanged.  Implementation notes:
//   * the file is mapped read-only and indexed once into a {type -> [extent]} table; there is no
//     second in-memory copy (reference quirk Q13), libzkhip uploads straight from the mapping;
//   * every cursor move is bounds-checked: a truncated file is an error, not a wild read;
//   * errors are C++ exceptions thrown BY VALUE with the reference's message texts (its
//     `throw new ...` escapes `catch (std::exception&)` and aborts — quirk Q1);
//   * primes are 32-byte little-endian arrays (no gmp on this path).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace BinFileUtils {

class BinFile {
public:
    BinFile(const std::string &fileName, const std::string &type, uint32_t maxVersion);
    // the same container held in memory (a .wtns image that arrived in a request body): the image is owned by the object
    BinFile(std::string &&image, const std::string &type, uint32_t maxVersion);
    ~BinFile();
    BinFile(const BinFile &) = delete;
    BinFile &operator=(const BinFile &) = delete;

    // sequential section reads
    void startReadSection(uint32_t sectionId, uint32_t sectionPos = 0);
    void endReadSection(bool check = true);
    uint32_t readU32LE();
    uint64_t readU64LE();
    void *read(uint64_t len);

    // random access
    void *getSectionData(uint32_t sectionId, uint32_t sectionPos = 0);
    uint64_t getSectionSize(uint32_t sectionId, uint32_t sectionPos = 0);

private:
    struct Extent {
        uint64_t begin, length;
    };
    const Extent &extent(uint32_t id, uint32_t nth) const;
    const uint8_t *take(uint64_t len);   // advance the cursor, checked

    void indexSections(const std::string &type, uint32_t maxVersion);
    uint8_t *map_ = nullptr;
    uint64_t mapLen_ = 0;
    std::string owned_;                  // in-memory images (map_ points into it; nothing to unmap)
    uint64_t cursor_ = 0;
    std::map<uint32_t, std::vector<Extent>> index_;
    const Extent *open_ = nullptr;
};

std::unique_ptr<BinFile> openExisting(const std::string &filename, const std::string &type, uint32_t maxVersion);
std::unique_ptr<BinFile> fromMemory(std::string &&image, const std::string &type, uint32_t maxVersion);

}   // namespace BinFileUtils

namespace ZKeyUtils {

// zkey sections 1-2 + the record count of section 4 (src/zkey_utils.cpp:17-52)
class Header {
public:
    uint32_t n8q = 0;
    std::array<uint8_t, 32> qPrime{};
    uint32_t n8r = 0;
    std::array<uint8_t, 32> rPrime{};
    uint32_t nVars = 0, nPublic = 0, domainSize = 0;
    uint64_t nCoefs = 0;
    void *vk_alpha1 = nullptr, *vk_beta1 = nullptr, *vk_beta2 = nullptr;
    void *vk_gamma2 = nullptr, *vk_delta1 = nullptr, *vk_delta2 = nullptr;
};
std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f);

}   // namespace ZKeyUtils

namespace WtnsUtils {

// wtns section 1 (src/wtns_utils.cpp:12-25)
class Header {
public:
    uint32_t n8 = 0;
    std::array<uint8_t, 32> prime{};
    uint32_t nVars = 0;
};
std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f);

}   // namespace WtnsUtils
