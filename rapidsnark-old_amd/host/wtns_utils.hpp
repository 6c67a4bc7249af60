// WtnsUtils::loadHeader — reference src/wtns_utils.hpp:10-21, src/wtns_utils.cpp:12-25.
#pragma once
#include <array>
#include <cstdint>
#include <memory>

#include "binfile_utils.hpp"

namespace WtnsUtils {

class Header {
public:
    uint32_t n8 = 0;
    std::array<uint8_t, 32> prime{};
    uint32_t nVars = 0;
};

std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile *f);

}   // namespace WtnsUtils
