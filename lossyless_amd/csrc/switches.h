// The library's A/B switches as LINK-TIME functions (round 6).  Rounds 1-5 read them from the environment
// through a reader that the product build folded to nullptr -- every product translation unit carried the switch sites
// and the conditional regions around the code they select.  Now the product library links switches_product.cpp
// (constants: the library reads NO environment variable, tests/test_gpu_variants.py) and the tools/ builds (`make ablation`,
// `make probes`) link ablation/switches_env.cpp, which reads the LLA_* variables once per process.  The translation units that
// call these are the same object code in every build.
#pragma once

namespace lla {
namespace sw {

// ---- tower (tower.hip)
int zigzag();             // LLA_VIT_ZIGZAG: 1 = the tower's kernels walk the rows in alternating directions (GemmParams::rev)
bool prune_last_block();  // LLA_VIT_PRUNE_LAST: after the last block's attention only the class rows are computed
int default_chunk();      // LLA_VIT_CHUNK: images per library slice when the caller passes chunk <= 0 (8704)
int lane_split_min();     // LLA_VIT_SPLIT_MIN: batches below this many images stay on one lane (640)
int tower_lanes();        // LLA_VIT_STREAMS: 1 (product: two lanes are not bit-reproducible, docs/history/DESIGN_rounds_1-5.md 5.3) or 2
// ---- RN50-CLIP tower (rn50.hip)
bool rn50_fuse_downsample();   // LLA_RN50_FUSE_DS: conv3 + downsample of a stage's first block as one GEMM
bool rn50_direct_conv();       // LLA_RN50_DIRECT: narrow 3x3 convolutions on conv_direct.hip
bool rn50_im2col();            // LLA_RN50_IM2COL: 3x3 convolutions through an im2col matrix (A/B)
bool rn50_fused_bottleneck();  // LLA_RN50_FUSED_BLOCK: layer1 blocks 1 and 2 as one kernel each (bottleneck_fused.hip)
// ---- preprocess.hip
int preprocess_band_rows();    // LLA_PRE_TH: first band height the fused resize tries (28)

}  // namespace sw
}  // namespace lla
