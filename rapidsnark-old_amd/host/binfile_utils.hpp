// Compatibility include: the reference's binfile_utils.hpp surface lives in zkfile.hpp.
#pragma once
#include "zkfile.hpp"
