// BinFileUtils — sectioned binary container used by .zkey / .wtns files.
// Same public surface as the reference (src/binfile_utils.hpp:10-52): openExisting,
// startReadSection/endReadSection, getSectionData/getSectionSize, readU32LE/readU64LE/read.
// Differences, on purpose:
//   Q1  errors are thrown BY VALUE (the reference throws pointers that escape
//       catch(std::exception&) and abort); message texts are kept.
//   Q13 the file is mapped read-only (mmap) instead of mmap + malloc + memcpy of the whole
//       image; libzkhip copies the sections to the GPU and the mapping is dropped afterwards.
// Bounds are checked so a truncated file is an error, not an out-of-bounds read.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace BinFileUtils {

class BinFile {
public:
    BinFile(const std::string &fileName, const std::string &type, uint32_t maxVersion);
    ~BinFile();
    BinFile(const BinFile &) = delete;
    BinFile &operator=(const BinFile &) = delete;

    void startReadSection(uint32_t sectionId, uint32_t sectionPos = 0);
    void endReadSection(bool check = true);

    void *getSectionData(uint32_t sectionId, uint32_t sectionPos = 0);
    uint64_t getSectionSize(uint32_t sectionId, uint32_t sectionPos = 0);

    uint32_t readU32LE();
    uint64_t readU64LE();
    void *read(uint64_t len);

private:
    struct Section {
        uint64_t offset, size;
    };
    const Section &find(uint32_t sectionId, uint32_t sectionPos) const;
    void need(uint64_t len) const;

    uint8_t *base_ = nullptr;
    uint64_t size_ = 0;
    uint64_t pos_ = 0;
    std::map<uint32_t, std::vector<Section>> sections_;
    const Section *reading_ = nullptr;
    std::string type_;
    uint32_t version_ = 0;
};

std::unique_ptr<BinFile> openExisting(const std::string &filename, const std::string &type, uint32_t maxVersion);

}   // namespace BinFileUtils
