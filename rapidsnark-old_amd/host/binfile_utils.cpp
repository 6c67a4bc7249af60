#include "binfile_utils.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <system_error>
#include <unistd.h>

namespace BinFileUtils {

BinFile::BinFile(const std::string &fileName, const std::string &type, uint32_t maxVersion) {
    int fd = open(fileName.c_str(), O_RDONLY);
    if (fd == -1) throw std::system_error(errno, std::generic_category(), "open");
    struct stat sb;
    if (fstat(fd, &sb) == -1) {
        int e = errno;
        close(fd);
        throw std::system_error(e, std::generic_category(), "fstat");
    }
    size_ = (uint64_t)sb.st_size;
    if (size_ > 0) {
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            int e = errno;
            close(fd);
            throw std::system_error(e, std::generic_category(), "mmap");
        }
        base_ = static_cast<uint8_t *>(m);
    }
    close(fd);

    need(4);
    type_.assign(reinterpret_cast<const char *>(base_), 4);
    pos_ = 4;
    if (type_ != type) throw std::invalid_argument("Invalid file type. It should be " + type + " and it us " + type_);
    version_ = readU32LE();
    if (version_ > maxVersion)
        throw std::invalid_argument("Invalid version. It should be <=" + std::to_string(maxVersion) + " and it us " +
                                    std::to_string(version_));
    uint32_t nSections = readU32LE();
    for (uint32_t i = 0; i < nSections; i++) {
        uint32_t sType = readU32LE();
        uint64_t sSize = readU64LE();
        need(sSize);
        sections_[sType].push_back(Section{pos_, sSize});
        pos_ += sSize;
    }
    pos_ = 0;
}

BinFile::~BinFile() {
    if (base_) munmap(base_, size_);
}

void BinFile::need(uint64_t len) const {
    if (len > size_ || pos_ > size_ - len) throw std::range_error("Unexpected end of file");
}

const BinFile::Section &BinFile::find(uint32_t sectionId, uint32_t sectionPos) const {
    auto it = sections_.find(sectionId);
    if (it == sections_.end()) throw std::range_error("Section does not exist: " + std::to_string(sectionId));
    if (sectionPos >= it->second.size())
        throw std::range_error("Section pos too big. There are " + std::to_string(it->second.size()) +
                               " and it's trying to access section: " + std::to_string(sectionPos));
    return it->second[sectionPos];
}

void BinFile::startReadSection(uint32_t sectionId, uint32_t sectionPos) {
    const Section &s = find(sectionId, sectionPos);
    if (reading_ != nullptr) throw std::range_error("Already reading a section");
    pos_ = s.offset;
    reading_ = &s;
}

void BinFile::endReadSection(bool check) {
    if (check && reading_ && pos_ - reading_->offset != reading_->size) {
        reading_ = nullptr;
        throw std::range_error("Invalid section size");
    }
    reading_ = nullptr;
}

void *BinFile::getSectionData(uint32_t sectionId, uint32_t sectionPos) { return base_ + find(sectionId, sectionPos).offset; }

uint64_t BinFile::getSectionSize(uint32_t sectionId, uint32_t sectionPos) { return find(sectionId, sectionPos).size; }

uint32_t BinFile::readU32LE() {
    need(4);
    uint32_t v;
    memcpy(&v, base_ + pos_, 4);
    pos_ += 4;
    return v;
}

uint64_t BinFile::readU64LE() {
    need(8);
    uint64_t v;
    memcpy(&v, base_ + pos_, 8);
    pos_ += 8;
    return v;
}

void *BinFile::read(uint64_t len) {
    need(len);
    void *p = base_ + pos_;
    pos_ += len;
    return p;
}

std::unique_ptr<BinFile> openExisting(const std::string &filename, const std::string &type, uint32_t maxVersion) {
    return std::unique_ptr<BinFile>(new BinFile(filename, type, maxVersion));
}

}   // namespace BinFileUtils
