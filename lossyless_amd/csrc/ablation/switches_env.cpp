// The tools/ builds' switch values (../switches.h): read once per process from the LLA_* environment variables, defaults =
// the product's constants (../switches_product.cpp).  Linked by `make ablation` / `make probes` / `make variant` INSTEAD of
// switches_product.cpp; never part of liblossyless_amd.so.
#include <cstdlib>

#include "../switches.h"

namespace lla {
namespace sw {
namespace {
int env_int(const char *name, int dflt) { const char *e = std::getenv(name); return e ? std::atoi(e) : dflt; }
bool env_not0(const char *name) { const char *e = std::getenv(name); return !(e && e[0] == '0'); }
bool env_is1(const char *name) { const char *e = std::getenv(name); return e && e[0] == '1'; }
}  // namespace

int zigzag() { static const int v = env_not0("LLA_VIT_ZIGZAG") ? 1 : 0; return v; }
bool prune_last_block() { static const bool v = env_not0("LLA_VIT_PRUNE_LAST"); return v; }
int default_chunk() { static const int v = [] { const int c = env_int("LLA_VIT_CHUNK", 0); return c > 0 ? c : 8704; }(); return v; }
int lane_split_min() { static const int v = [] { const int n = env_int("LLA_VIT_SPLIT_MIN", 640); return n >= 2 ? n : 2; }(); return v; }
int tower_lanes() { static const int v = env_int("LLA_VIT_STREAMS", 1) >= 2 ? 2 : 1; return v; }
bool rn50_fuse_downsample() { static const bool v = env_not0("LLA_RN50_FUSE_DS"); return v; }
bool rn50_direct_conv() { static const bool v = env_not0("LLA_RN50_DIRECT"); return v; }
bool rn50_im2col() { static const bool v = env_is1("LLA_RN50_IM2COL"); return v; }
bool rn50_fused_bottleneck() { static const bool v = env_not0("LLA_RN50_FUSED_BLOCK"); return v; }
int preprocess_band_rows() { static const int v = env_int("LLA_PRE_TH", 28); return v; }

}  // namespace sw
}  // namespace lla
