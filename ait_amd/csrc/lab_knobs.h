// ait_amd/csrc/lab_knobs.h -- every build-time experiment switch of the library, in ONE place.
//
// The shipped library is built WITHOUT AIT_LAB_KNOBS (ait_amd/build.py never defines it; tests/test_abi.py asserts that
// ait_lab_build() of the shipped .so returns 0): `Knobs` is then `Product`, every constant below has its product value and
// `if constexpr` drops the other branch -- the sources hold no other #if / #ifdef for experiments and read no environment.
// A lab build (scripts/build_variant.py <name> <sources> knob=value ...) compiles the named sources with
//   -DAIT_LAB_KNOBS='"<generated header>"'
// where the generated header defines   struct Knobs : Product { static constexpr <type> <knob> = <value>; ... };
// for same-box A/Bs; such a library is named libait_hip_<name>.so, lives under scripts/_lab/, and is never loaded by
// ait_amd/_lib.py.
#pragma once

namespace ait_lab {

struct Product {
  // gemm_bf16s.hip: keep every bf16-storage product on the 256 x 128 x 32 tile
  static constexpr bool bf16s_small_only = false;
  // gemm_bf16s.hip: the shortest reduction the 256 x 256 x 64 tile takes
  static constexpr int bf16s_big_min_k = 512;
  // gemm_bf16s.hip: weight gradients always through f32 atomics (never the stored K-range partials)
  static constexpr bool tn_atomics = false;
  // gemm_bf16s.hip: the 256 x 256 tile's under-filled last round of tiles cut along K (pieces first, a finishing launch behind)
  static constexpr bool bf16s_cut = true;
  // gemm_bf16s.hip: the weight-gradient kernel's items in one contiguous chunk per XCD (false: dealt over all workgroups)
  static constexpr bool tn_xcd_chunks = true;
  // gemm_p3.hip: the fewest 256 x 256 tiles the pre-split-weight kernel takes
  static constexpr int p3_min_tiles = 128;
  // mha_fused_bwd.hip: which of the attention tile's backward products run in the split form (1 dV, 2 dPd, 4 dQ, 8 dK)
  static constexpr int fb_split = 15;
  // mha_fused_bwd.hip: du = df fc_w on the f32 instruction
  static constexpr bool fb_du_f32 = false;
  // roi_align.hip: the plane-resident NCHW forward (1.05 ms against the window-staged kernel's 0.81)
  static constexpr bool roi_fwd_plane = false;
  // roi_align_nhwc.hip: the separable two-stage forward (profiles/r05_roi_align_sep.txt), cells a lane keeps in flight in it
  static constexpr bool roi_fwd_separable = false;
  static constexpr int roi_cells = 4;
  // tail.hip: the residual + gate epilogue on the 256 x 256 tile as well
  static constexpr bool tail_resg_p3 = false;
  // transformer.hip, the bf16 configuration's A/Bs: f32 q / k / v; the f32-storage projection backward; the f32-storage
  // feed-forward
  static constexpr bool no_bf16_qkv = false;
  static constexpr bool no_bf16_attn = false;
  static constexpr bool no_bf16_ffn = false;
  // transformer.hip, the bf16 configuration: dec_trans and its gradients on bf16 operands also with f32 `out` / `d_out`
  static constexpr bool dec_trans_bf16 = true;
  // transformer.hip, the bf16 configuration: d fc_w of the attention blocks on the bf16 weight-gradient kernel (false: the f32 product)
  static constexpr bool fcw_bf16 = true;
  // attn_impl.h: column-paired right-operand loads for f32 q / k / v as well; six MFMA terms whatever the operands are
  static constexpr bool f32_pairs = false;
  static constexpr bool no_bf16_planes = false;
  // tail.hip: layer4's activations in position-major row order (false: map-major, round 5's)
  static constexpr bool l4_pm = true;
  // conv_f32.hip: the 3x3 weight gradient of position-major maps over position blocks, out-of-map (position, tap) pairs skipped
  static constexpr bool l4_pm_wgrad = true;
  // conv_f32.hip: the 3x3 forward / data gradient of position-major maps with the out-of-map taps skipped (two-group stream-K cut)
  static constexpr bool l4_pm_skip = true;
  // tail.hip: layer4 on bf16 storage under AIT_CTX_BF16 (false: f32 tensors, operands rounded in registers -- round 5's)
  static constexpr bool tail_bf16s = true;
  // gemm_f32.hip: products of fewer than 512 256x128 tiles on the 128x128 split tile (0 off, 1 static work list, 2 with the
  // scheduler scratch: tickets + stream-K)
  static constexpr int mid_tile = 0;
  // gemm_f32_impl.h: round 4's last-round cost model; slab-times a stream-K cut of the last round must save
  static constexpr bool old_last_round = false;
  static constexpr double sk_pays = 12.0;
};

#ifdef AIT_LAB_KNOBS
#include AIT_LAB_KNOBS
constexpr int kLabBuild = 1;
#else
using Knobs = Product;
constexpr int kLabBuild = 0;
#endif

}  // namespace ait_lab
