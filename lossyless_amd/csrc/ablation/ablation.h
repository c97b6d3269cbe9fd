// tools/ builds only (make ablation / make probes / make variant): the environment reader of the A/B switches.  The product
// library has no such function -- its translation units contain no switch site (switches.h, gemm_pp.hip).
#pragma once
#include <cstdlib>

namespace lla {
inline const char *lla_getenv(const char *name) { return std::getenv(name); }
}  // namespace lla
