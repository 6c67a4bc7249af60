// ZKeyUtils::loadHeader — reference src/zkey_utils.hpp:11-36, src/zkey_utils.cpp:17-52.
// Same field names.  The primes are 32-byte little-endian arrays instead of mpz_t: nothing
// on this path needs gmp (the GPU box is not guaranteed to have it).
#pragma once
#include <array>
#include <cstdint>
#include <memory>

#include "binfile_utils.hpp"

namespace ZKeyUtils {

class Header {
public:
    uint32_t n8q = 0;
    std::array<uint8_t, 32> qPrime{};
    uint32_t n8r = 0;
    std::array<uint8_t, 32> rPrime{};

    uint32_t nVars = 0;
    uint32_t nPublic = 0;
    uint32_t domainSize = 0;
    uint64_t nCoefs = 0;

    void *vk_alpha1 = nullptr;
    void *vk_beta1 = nullptr;
    void *vk_beta2 = nullptr;
    void *vk_gamma2 = nullptr;
    void *vk_delta1 = nullptr;
    void *vk_delta2 = nullptr;
};

std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f);

}   // namespace ZKeyUtils
