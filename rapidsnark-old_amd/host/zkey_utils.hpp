// Compatibility include: the reference's zkey_utils.hpp surface lives in zkfile.hpp.
#pragma once
#include "zkfile.hpp"
