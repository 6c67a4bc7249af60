// Compatibility include: the reference's wtns_utils.hpp surface lives in zkfile.hpp.
#pragma once
#include "zkfile.hpp"
