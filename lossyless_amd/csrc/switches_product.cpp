// The product library's switch values (switches.h): constants.  liblossyless_amd.so reads no environment variable.
#include "switches.h"

namespace lla {
namespace sw {

int zigzag() { return 1; }
bool prune_last_block() { return true; }
// 8704 images = 1700 row tiles of 256 rows: the persistent GEMMs' rounds come out full on 256 CUs (19.9 / 59.8 / 79.7 rounds);
// tower alone 94.9k img/s at 1024, 99.5k at 4352, 101.0k at 8704 (tools/slice_probe.py, round 4)
int default_chunk() { return 8704; }
int lane_split_min() { return 640; }
int tower_lanes() { return 1; }
bool rn50_fuse_downsample() { return true; }
bool rn50_direct_conv() { return true; }
bool rn50_im2col() { return false; }
bool rn50_fused_bottleneck() { return true; }
int preprocess_band_rows() { return 28; }

}  // namespace sw
}  // namespace lla
