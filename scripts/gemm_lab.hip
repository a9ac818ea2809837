// scripts/gemm_lab.hip -- measuring harness for the fp32 MFMA GEMM (NOT part of the product library).
// Stand-alone executable (no Python on the GPU box):
//     scripts/build_gemm_lab.sh            (here: cross-compiles for gfx950)
//     scripts/_gemm_lab ab [rounds]        interleaved A/B of the round-1 kernel and the current one
//     scripts/_gemm_lab probe              one stamped launch per shape: workgroup time line, share of
//                                          the slab loop spent in the vmcnt wait / the barrier, clock
//     scripts/_gemm_lab pmc <shape> <variant> [n]   n launches of one variant (for rocprofv3 --pmc)
// The round-1 kernel comes from the history (81edb57), extracted by the build script into
// /tmp/ait_old_gemm_f32_impl.h with its namespace renamed; without it (-DNO_OLD) only the current one runs.
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../ait_amd/csrc/gemm_f32_impl.h"
#include "../ait_amd/csrc/p3_impl.h"
#ifndef NO_OLD
#include "/tmp/ait_old_gemm_f32_impl.h"
#endif

using namespace ait_gemm;

using NewD = Cfg<256, 128, 16, 4, 2, 4, MODE_DLDS>;
using NewD4 = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS>;
// sweep variants (all produce the same results)
using V_burst = Cfg<256, 128, 16, 4, 2, 4, MODE_DLDS, 3, KNOB_BURST>;
using V_prio = Cfg<256, 128, 16, 4, 2, 4, MODE_DLDS, 3, KNOB_PRIO>;
using V_ring4 = Cfg<256, 128, 16, 4, 2, 2, MODE_DLDS, 4, 0>;             // 96 KB: one workgroup per CU
using V_ring4b = Cfg<256, 128, 16, 4, 2, 2, MODE_DLDS, 4, KNOB_BURST>;
using V_d4 = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, 0>;                // 4 waves of 128x64
using V_d4b = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_BURST>;
using V_128 = Cfg<128, 128, 16, 2, 2, 2, MODE_DLDS, 3, 0>;               // 48 KB: three workgroups per CU
using V_128r4 = Cfg<128, 128, 16, 2, 2, 2, MODE_DLDS, 4, 0>;             // 64 KB: two
using V_sp8 = Cfg<256, 128, 16, 4, 2, 4, MODE_DLDS, 3, KNOB_SPREAD>;    // 8 waves, LDS reads spread over the MFMA steps
using V_sp4 = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPREAD>;    // 4 waves, same
using V_256sq = Cfg<256, 256, 16, 4, 4, 4, MODE_DLDS, 3, 0>;             // 16 waves, 96 KB: one workgroup per CU
using V_split = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT>;     // the product tile, products on the bf16 matrix pipe (3-way split, 6 terms)
using V_split8 = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_SPLIT_SIMPLE>;    // every split in front of its tile's MFMAs
using V_splitsq = Cfg<256, 256, 16, 2, 4, 2, MODE_DLDS, 3, KNOB_SPLIT>;   // 256x256, EIGHT waves of 128x64, one workgroup per CU (98 KB)
using V_rne = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_RNE>;      // the product tile with the planes rounded to nearest
using V_bf16 = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BF16>;      // operands rounded to bf16, one MFMA per block
using V_bp3 = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BP3>;    // B pre-split (P3), 8 waves of 64x128, one workgroup per CU
using V_bp3r = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BP3 | KNOB_RNE>;
using V_ab3 = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BP3 | KNOB_AP3 | KNOB_RNE>;   // both operands pre-split: no vector work
// (variants 27 / 28, two scaled fp16 planes per value and three MFMAs per block, were measured in round 4 --
//  profiles/r04_gemm_lab_f16x2.txt -- and removed with the knob in round 5: 22 bits under a tensor scale is not an f32 product)
using V_bp3p = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_COOP | KNOB_NOTICKET | KNOB_RNE>;     // every value split once per workgroup (LDS plane image)
using V_bp3n4 = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 4, KNOB_SPLIT | KNOB_BP3 | KNOB_NOTICKET>;
using V_bp3n4p = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 4, KNOB_SPLIT | KNOB_BP3 | KNOB_NOTICKET | KNOB_PRIO>;
using V_stag = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPREAD | KNOB_STAGGER>;   // the product tile, odd threadgroup slots start half a tile late
#ifndef NO_OLD
using OldD = ait_gemm_old::Cfg<256, 128, 16, 4, 2, 2, 6 + 256>;
using OldD4 = ait_gemm_old::Cfg<256, 128, 16, 2, 2, 2, 6 + 256>;
#endif

#define LAB_RES (1 << 20)      // lab-only flag: pass a residual operand
#define LAB_ALIAS_A (1 << 21)
#define LAB_ALIAS_B (1 << 22)
struct Shape {
  const char* name;
  int M, N, K, ta, tb, sk, flags;   // flags: AIT_GEMM_RELU (with a bias) | AIT_GEMM_ATOMIC is implied by sk > 1
};
static const Shape SHAPES[] = {
    {"qkv   NT", 76800, 1536, 512, 0, 1, 1, 0},
    {"ffn1  NT", 76800, 2048, 512, 0, 1, 1, AIT_GEMM_RELU},
    {"ffn2  NT", 76800, 512, 2048, 0, 1, 1, 0},
    {"dgrad NN", 76800, 512, 2048, 0, 0, 1, 0},
    {"dgrad NN", 76800, 2048, 512, 0, 0, 1, 0},
    {"wgrad TN", 512, 2048, 76800, 1, 0, 16, 0},
    {"wgrad TN", 2048, 512, 76800, 1, 0, 16, 0},
    {"wgrad TN", 1536, 512, 76800, 1, 0, 24, 0},
    {"qkv6r NT", 65536, 1536, 512, 0, 1, 1, 0},      // 3072 tiles = 6 full rounds of 512 slots
    {"qkvK4 NT", 76800, 1536, 4096, 0, 1, 1, 0},     // same tiles, 8x the slabs per tile
    {"l4c1  NT", 19200, 512, 2048, 0, 1, 1, AIT_GEMM_RELU},
    {"l4c3  NT", 19200, 2048, 512, 0, 1, 1, AIT_GEMM_RELU},
    {"dx+r  NN", 76800, 512, 2048, 0, 0, 1, LAB_RES},                      // dgrad + residual-gradient add
    {"dhmsk NN", 76800, 2048, 512, 0, 0, 1, LAB_RES | AIT_GEMM_MASK_POS},  // dgrad gated by the saved ReLU
    {"fc    NT", 76800, 512, 64, 0, 1, 1, 0},                              // K = 64: four slabs per tile
    {"trans NT", 76800, 1024, 512, 0, 1, 1, AIT_GEMM_RELU},                // dec_trans shape (+ bias)
    {"xq    NT", 76800, 512, 512, 0, 1, 1, 0},                             // cross-attention query projection: 600 tiles of 32 slabs
    {"l4c3r NT", 19328, 2048, 512, 0, 1, 1, AIT_GEMM_RELU | LAB_RES},      // layer4 conv3 + shift + shortcut + ReLU
    {"l4w2  TN", 512, 4608, 19328, 1, 0, 7, 0},                            // layer4 conv2's weight gradient as a plain product
    {"l4w2  TN", 512, 4608, 19328, 1, 0, 8, 0},
    {"rpnw  TN", 512, 9216, 9576, 1, 0, 8, 0},                             // the RPN convolution's weight gradient as a plain product
    // timing-only experiments (results meaningless): operand rows aliased onto one row (row pitch 0), so that
    // the loads are served from L1 / L2 whatever the tile -- what the kernel does with the memory system taken away
    {"qkv aA NT", 76800, 1536, 512, 0, 1, 1, LAB_ALIAS_A},
    {"qkv aB NT", 76800, 1536, 512, 0, 1, 1, LAB_ALIAS_B},
    {"qkv aAB  ", 76800, 1536, 512, 0, 1, 1, LAB_ALIAS_A | LAB_ALIAS_B},
    {"ffn2 aAB ", 76800, 512, 2048, 0, 1, 1, LAB_ALIAS_A | LAB_ALIAS_B},
    // tile counts that leave a badly filled last round (stream-K cases)
    {"ffn49 NT", 58800, 2048, 512, 0, 1, 1, AIT_GEMM_RELU},               // 49-row sequences: 230 x 16 tiles
    {"kv49  NT", 58800, 1024, 512, 0, 1, 1, 0},
    {"l4c2  NT", 19200, 512, 4608, 0, 1, 1, AIT_GEMM_RELU},               // layer4 3x3 as a plain GEMM
    {"l4dx  NN", 19200, 2048, 512, 0, 0, 1, LAB_RES},
    {"coatt NT", 9576, 512, 1024, 0, 1, 1, 0},
    // layer4 weight gradients at one K-range per XCD (their last round is cut evenly without scratch)
    {"l4w2  TN", 512, 4608, 19328, 1, 0, 8, 0},
    {"rpnw  TN", 512, 9216, 9576, 1, 0, 8, 0},                             // the RPN convolution's weight gradient as a plain product
    {"l4w3  TN", 2048, 512, 19328, 1, 0, 8, 0},
    {"l4w1  TN", 512, 2048, 19328, 1, 0, 8, 0},
    {"qkvw8 TN", 1536, 512, 76800, 1, 0, 8, 0},
};
static const int NSHAPES = sizeof(SHAPES) / sizeof(SHAPES[0]);

__global__ void scale_kernel(float* p, size_t n, float s) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] *= s;
}
__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (float)(int)x * (1.0f / 2147483648.0f);      // uniform [-1, 1)
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Problem {
  Shape s;
  float *A, *B, *C, *C2, *bias, *res;
  unsigned short* Bp3;      // B as P3 [N][K / 8][3][8] (variants 20, 21)
  unsigned short* Ap3;      // A as P3 [M][K / 8][3][8] (variant 26)
  unsigned long long* probe;
  GemmArgs g;
};

static void setup(Problem& p, const Shape& s) {
  p.s = s;
  const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
  CK(hipMalloc(&p.A, na * 4)); CK(hipMalloc(&p.B, nb * 4)); CK(hipMalloc(&p.C, nc * 4)); CK(hipMalloc(&p.C2, nc * 4));
  CK(hipMalloc(&p.bias, (size_t)s.N * 4));
  CK(hipMalloc(&p.res, nc * 4));
  fill_kernel<<<2048, 256>>>(p.res, nc, 0x777u);
  CK(hipMalloc(&p.probe, 4096 * AIT_PROBE_WORDS * 8));
  fill_kernel<<<2048, 256>>>(p.A, na, 0x1234567u);
  fill_kernel<<<2048, 256>>>(p.B, nb, 0x89abcdeu);
  fill_kernel<<<64, 256>>>(p.bias, (size_t)s.N, 0x5555u);
  if (const char* e = getenv("LAB_DATA_SCALE")) {      // operands multiplied by 2^-e: how the lab fp16 form behaves on small values
    const float sc = ldexpf(1.f, -atoi(e));
    scale_kernel<<<2048, 256>>>(p.A, na, sc);
    scale_kernel<<<2048, 256>>>(p.B, nb, sc);
  }
  CK(hipMemset(p.C, 0, nc * 4)); CK(hipMemset(p.C2, 0, nc * 4));
  p.Bp3 = nullptr;
  p.Ap3 = nullptr;
  if (!s.ta && s.K % 16 == 0) {
    CK(hipMalloc(&p.Bp3, nb * 6));
    ait_p3::Jobs jobs;
    jobs.n = 1;
    jobs.j[0] = s.tb ? ait_p3::Job{p.B, p.Bp3, s.N, s.K, s.K, 0, 0} : ait_p3::Job{p.B, p.Bp3, s.K, s.N, s.N, 1, 0};
    int rc = ait_p3::launch_split(jobs, 0);
    if (rc) { printf("p3 split rc %d\n", rc); exit(1); }
    CK(hipMalloc(&p.Ap3, na * 6));
    jobs.j[0] = ait_p3::Job{p.A, p.Ap3, s.M, s.K, s.K, 0, 0};
    rc = ait_p3::launch_split(jobs, 0);
    if (rc) { printf("p3 split (A) rc %d\n", rc); exit(1); }
  }
  CK(hipDeviceSynchronize());
  const int lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
  const int flags = (s.flags & ~(LAB_RES | LAB_ALIAS_A | LAB_ALIAS_B)) | (s.sk > 1 ? AIT_GEMM_ATOMIC : 0);
  int rc = make_args(s.ta, s.tb, s.M, s.N, s.K, 1.f, p.A, lda, p.B, ldb, p.C, s.N,
                     (s.flags & AIT_GEMM_RELU) ? p.bias : nullptr, (s.flags & LAB_RES) ? p.res : nullptr, flags, s.sk, 0,
                     0, 16, p.g);
  if (rc) { printf("make_args rc %d\n", rc); exit(1); }
  if (s.flags & LAB_ALIAS_A) p.g.lda = 0;
  if (s.flags & LAB_ALIAS_B) p.g.ldb = 0;
}
static void teardown(Problem& p) {
  if (p.Bp3) hipFree(p.Bp3);
  if (p.Ap3) hipFree(p.Ap3);
  hipFree(p.A); hipFree(p.B); hipFree(p.C); hipFree(p.C2); hipFree(p.bias); hipFree(p.res); hipFree(p.probe);
}

// variant 0: round-1 kernel; 1: current kernel; 2: current kernel with the stamp probe
// the harness owns the scheduler workspace it hands to the launches (g_use_ws = false: whole tiles, static lists)
static SchedWs g_ws;
static bool g_use_ws = true;
static SchedWs lab_ws() {
  if (!g_ws.p) {
    g_ws.bytes = kCtlBytes + (size_t)kMaxSlots * 128 * 128 * sizeof(float);      // covers every variant's tile x slots
    CK(hipMalloc(&g_ws.p, g_ws.bytes));
    CK(hipMemset(g_ws.p, 0, kCtlBytes));
  }
  return g_use_ws ? g_ws : SchedWs();
}
template <class T, bool AK, bool BKC, class Probe>
static int run_epi(const GemmArgs& g, int slots) {
  if (g.flags & AIT_GEMM_ATOMIC) return launch<T, AK, BKC, EPI_ATOMIC, Probe>(g, 0, lab_ws(), slots);
  if (g.residual) return launch<T, AK, BKC, EPI_RES, Probe>(g, 0, lab_ws(), slots);
  return launch<T, AK, BKC, EPI_STORE, Probe>(g, 0, lab_ws(), slots);
}
template <class T>
static int run_tile(const GemmArgs& g, bool ak, bool bk, int slots) {
  if (!ak && !bk) return run_epi<T, false, false, NoProbe>(g, slots);
  if (ak && bk) return run_epi<T, true, true, NoProbe>(g, slots);
  if (ak && !bk) return run_epi<T, true, false, NoProbe>(g, slots);
  return run_epi<T, false, true, NoProbe>(g, slots);
}
static const char* VNAMES[] = {"old", "new", "probe", "burst", "prio", "ring4", "ring4+burst", "4waves", "4waves+burst",
                               "128sq", "128sq ring4", "256sq 16w", "8w spread", "4w spread", "4w spr noSK", "4w spr stag", "split 4w", "split simple", "split 256sq", "split rne", "bp3 256sq", "bp3 rne", "bf16 x1", "coop 256sq", "bp3 ring4", "bp3 ring4 prio", "ap3+bp3"};
template <class Probe>
static int run_new(const GemmArgs& g, bool ak, bool bk, int slots) {
  if (!ak && !bk) return run_epi<NewD4, false, false, Probe>(g, slots);
  if (ak && bk) return run_epi<NewD, true, true, Probe>(g, slots);
  if (ak && !bk) return run_epi<NewD, true, false, Probe>(g, slots);
  return run_epi<NewD, false, true, Probe>(g, slots);
}
#ifndef NO_OLD
static int run_old(const GemmArgs& gn, bool ak, bool bk) {
  ait_gemm_old::GemmArgs g;
  g.A = gn.A; g.B = gn.B; g.C = gn.C; g.bias = gn.bias; g.residual = gn.residual;
  g.M = gn.M; g.N = gn.N; g.K = gn.K; g.lda = gn.lda; g.ldb = gn.ldb; g.ldc = gn.ldc;
  g.c_colblk = gn.c_colblk; g.c_batch = gn.c_batch; g.alpha = gn.alpha; g.flags = gn.flags;
  g.k_per_split = gn.k_per_split; g.splits = gn.splits;
  if (!ak && !bk) return ait_gemm_old::dispatch<OldD4>(g, false, false, 0);
  return ait_gemm_old::dispatch<OldD>(g, ak, bk, 0);
}
#endif
static int g_slots = 0;
static int run(Problem& p, int variant, float* out) {
  GemmArgs g = p.g;
  g.C = out;
  const bool ak = !p.s.ta, bk = p.s.tb != 0;
  if (variant == 1) {                // the reference for the sweep's result check: whole tiles only
    g_use_ws = false;
    const int rc = run_new<NoProbe>(g, ak, bk, g_slots);
    g_use_ws = true;
    return rc;
  }
  if (variant == 2) {
    g.probe = p.probe;
    if (getenv("LAB_PROBE_COOP")) {           // the stamp probe on the cooperative-split tile
      if (!ak && !bk) return run_epi<V_bp3p, false, false, StampProbe>(g, g_slots);
      if (ak && bk) return run_epi<V_bp3p, true, true, StampProbe>(g, g_slots);
      if (ak && !bk) return run_epi<V_bp3p, true, false, StampProbe>(g, g_slots);
      return run_epi<V_bp3p, false, true, StampProbe>(g, g_slots);
    }
    if (getenv("LAB_PROBE_BP3") && p.Bp3) {    // the stamp probe on the pre-split-B tile
      g.B = reinterpret_cast<const float*>(p.Bp3);
      g.ldb = p.s.K / 2 * 3;
      if (atoi(getenv("LAB_PROBE_BP3")) == 26) {
        g.A = reinterpret_cast<const float*>(p.Ap3);
        g.lda = p.s.K / 2 * 3;
        return run_epi<V_ab3, true, true, StampProbe>(g, g_slots);
      }
      if (atoi(getenv("LAB_PROBE_BP3")) == 21) return run_epi<V_bp3r, true, true, StampProbe>(g, g_slots);
      if (atoi(getenv("LAB_PROBE_BP3")) == 4) return run_epi<V_bp3n4, true, true, StampProbe>(g, g_slots);
      return run_epi<V_bp3, true, true, StampProbe>(g, g_slots);
    }
    if (getenv("LAB_PROBE_SPLIT")) {      // the stamp probe on the split tile
      if (!ak && !bk) return run_epi<V_split, false, false, StampProbe>(g, g_slots);
      if (ak && bk) return run_epi<V_split, true, true, StampProbe>(g, g_slots);
      if (ak && !bk) return run_epi<V_split, true, false, StampProbe>(g, g_slots);
      return run_epi<V_split, false, true, StampProbe>(g, g_slots);
    }
    return run_new<StampProbe>(g, ak, bk, g_slots);
  }
  switch (variant) {
    case 3: return run_tile<V_burst>(g, ak, bk, g_slots);
    case 4: return run_tile<V_prio>(g, ak, bk, g_slots);
    case 5: return run_tile<V_ring4>(g, ak, bk, g_slots);
    case 6: return run_tile<V_ring4b>(g, ak, bk, g_slots);
    case 7: return run_tile<V_d4>(g, ak, bk, g_slots);
    case 8: return run_tile<V_d4b>(g, ak, bk, g_slots);
    case 9: return run_tile<V_128>(g, ak, bk, g_slots);
    case 10: return run_tile<V_128r4>(g, ak, bk, g_slots);
    case 11: return run_tile<V_256sq>(g, ak, bk, g_slots);
    case 12: return run_tile<V_sp8>(g, ak, bk, g_slots);
    case 13: return run_tile<V_sp4>(g, ak, bk, g_slots);
    case 14: {                      // the same kernel without the stream-K work list
      g_use_ws = false;
      const int rc = run_tile<V_sp4>(g, ak, bk, g_slots);
      g_use_ws = true;
      return rc;
    }
    case 15: return run_tile<V_stag>(g, ak, bk, g_slots);
    case 16: return run_tile<V_split>(g, ak, bk, g_slots);
    case 17: return run_tile<V_split8>(g, ak, bk, g_slots);
    case 18: return run_tile<V_splitsq>(g, ak, bk, g_slots);
    case 19: return run_tile<V_rne>(g, ak, bk, g_slots);
    case 22: return run_tile<V_bf16>(g, ak, bk, g_slots);
    case 23: return run_tile<V_bp3p>(g, ak, bk, g_slots);
    case 20: case 21: case 24: case 25: case 26: {
      if (!p.Bp3 || (p.s.flags & (LAB_ALIAS_A | LAB_ALIAS_B))) return -99;
      g.B = reinterpret_cast<const float*>(p.Bp3);
      g.ldb = p.s.K / 2 * 3;
      if (variant == 26) {
        g.A = reinterpret_cast<const float*>(p.Ap3);
        g.lda = p.s.K / 2 * 3;
        return run_epi<V_ab3, true, true, NoProbe>(g, g_slots);
      }
      if (variant == 21) return run_epi<V_bp3r, true, true, NoProbe>(g, g_slots);
      if (variant == 23) return -99;
      if (variant == 24) return run_epi<V_bp3n4, true, true, NoProbe>(g, g_slots);
      if (variant == 25) return run_epi<V_bp3n4p, true, true, NoProbe>(g, g_slots);
      return run_epi<V_bp3, true, true, NoProbe>(g, g_slots);
    }
    default: break;
  }
#ifndef NO_OLD
  return run_old(g, ak, bk);
#else
  return -99;
#endif
}

static double tflops(const Shape& s, double ms) { return 2.0 * s.M * s.N * s.K / ms / 1e9; }

static float time_launches(Problem& p, int variant, float* out, int n) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t nc = (size_t)p.s.M * p.s.N;
  if (p.s.sk > 1) CK(hipMemsetAsync(out, 0, nc * 4, 0));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; i++) run(p, variant, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / n;
}

static double max_diff(Problem& p, int n_launch_c2) {
  const size_t nc = (size_t)p.s.M * p.s.N;
  std::vector<float> a(nc), b(nc);
  CK(hipMemcpy(a.data(), p.C, nc * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), p.C2, nc * 4, hipMemcpyDeviceToHost));
  double m = 0, ref = 0;
  for (size_t i = 0; i < nc; i++) {
    m = std::max(m, (double)fabsf(a[i] - b[i]));
    ref = std::max(ref, (double)fabsf(a[i]));
  }
  (void)n_launch_c2;
  return ref > 0 ? m / ref : m;
}

static void mode_ab(int rounds, int first, int last) {
  printf("%-9s %6s %5s %6s | %8s %8s | %8s %8s | %7s | rel.diff\n", "shape", "M", "N", "K", "old med", "old max", "new med", "new max", "new/old");
  for (int si = first; si < last; si++) {
    Problem p;
    setup(p, SHAPES[si]);
    const size_t nc = (size_t)p.s.M * p.s.N;
    // correctness: one launch of each into zeroed outputs
    CK(hipMemset(p.C, 0, nc * 4)); CK(hipMemset(p.C2, 0, nc * 4));
    int rc0 = run(p, 0, p.C), rc1 = run(p, 1, p.C2);
    CK(hipDeviceSynchronize());
    const double diff = (rc0 == 0 && rc1 == 0) ? max_diff(p, 1) : -1;
    for (int w = 0; w < 2; w++) { time_launches(p, 0, p.C, 3); time_launches(p, 1, p.C2, 3); }
    std::vector<double> t0, t1;
    for (int r = 0; r < rounds; r++) {
      t0.push_back(tflops(p.s, time_launches(p, 0, p.C, 10)));
      t1.push_back(tflops(p.s, time_launches(p, 1, p.C2, 10)));
    }
    std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end());
    printf("%-9s %6d %5d %6d | %8.1f %8.1f | %8.1f %8.1f | %7.3f | %.2e (rc %d %d)\n", p.s.name, p.s.M, p.s.N, p.s.K,
           t0[rounds / 2], t0[rounds - 1], t1[rounds / 2], t1[rounds - 1], t1[rounds / 2] / t0[rounds / 2], diff, rc0, rc1);
    fflush(stdout);
    teardown(p);
  }
}

// every variant on every shape, interleaved rounds in one process; median TF/s
static void mode_sweep(int rounds, int first, int last) {
  int vs[24] = {0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13};
  int nv = 13;
  if (const char* e = getenv("LAB_VARIANTS")) {       // e.g. LAB_VARIANTS=1,7,12,13
    nv = 0;
    for (const char* q = e; *q && nv < 24;) { vs[nv++] = atoi(q); q = strchr(q, ','); if (!q) break; q++; }
  }
  printf("%-9s %6s %5s %6s |", "shape", "M", "N", "K");
  for (int i = 0; i < nv; i++) printf(" %12s", VNAMES[vs[i]]);
  printf("\n");
  for (int si = first; si < last; si++) {
    Problem p;
    setup(p, SHAPES[si]);
    const size_t nc = (size_t)p.s.M * p.s.N;
    std::vector<std::vector<double>> t(nv);
    std::vector<int> ok(nv, 1);
    CK(hipMemset(p.C, 0, nc * 4));
    run(p, 1, p.C);
    for (int i = 0; i < nv; i++) {       // correctness of every variant against the current kernel
      CK(hipMemset(p.C2, 0, nc * 4));
      const int rc = run(p, vs[i], p.C2);
      CK(hipDeviceSynchronize());
      const double md = rc == 0 ? max_diff(p, 1) : -1.0;
      if (getenv("LAB_PRINT_DIFF")) fprintf(stderr, "  %s / %s: max |diff| / max |ref| = %.3g\n", p.s.name, VNAMES[vs[i]], md);
      if (rc != 0 || md > (vs[i] == 22 ? 3e-2 : 2e-5)) { ok[i] = 0; fprintf(stderr, "  %s / %s: rc %d, max |diff| / max |ref| = %.3g\n", p.s.name, VNAMES[vs[i]], rc, md); }      // stream-K and split-K change the summation order
      if (rc == 0) time_launches(p, vs[i], p.C2, 3);
    }
    for (int r = 0; r < rounds; r++)
      for (int i = 0; i < nv; i++) t[i].push_back(tflops(p.s, time_launches(p, vs[i], p.C2, 8)));
    printf("%-9s %6d %5d %6d |", p.s.name, p.s.M, p.s.N, p.s.K);
    for (int i = 0; i < nv; i++) {
      std::sort(t[i].begin(), t[i].end());
      printf(" %10.1f%s", t[i][rounds / 2], ok[i] ? "  " : " X");
    }
    printf("\n");
    fflush(stdout);
    teardown(p);
  }
}

// The same launch queued continuously for `seconds` (64 launches per timed window, windows back to back
// with no host gap): does the rate hold once the chip is in its sustained power state?  (The sweep times
// bursts of 8 launches between host synchronisations.)
static void mode_sustain(int variant, double seconds, int si) {
  Problem p;
  setup(p, SHAPES[si]);
  const int per = 64, nwin = 256;
  std::vector<hipEvent_t> ev(nwin + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  const float one = time_launches(p, variant, p.C, 8);
  int windows = (int)(seconds * 1000.0 / (one * per)) + 1;
  if (windows > nwin) windows = nwin;
  CK(hipEventRecord(ev[0], 0));
  for (int w = 0; w < windows; w++) {
    for (int i = 0; i < per; i++) run(p, variant, p.C);
    CK(hipEventRecord(ev[w + 1], 0));
  }
  CK(hipEventSynchronize(ev[windows]));
  printf("%s variant %s: burst of 8 = %.1f TF; sustained windows of %d launches:\n", p.s.name, VNAMES[variant], tflops(p.s, one), per);
  for (int w = 0; w < windows; w++) {
    float ms;
    CK(hipEventElapsedTime(&ms, ev[w], ev[w + 1]));
    printf(" %.1f", tflops(p.s, ms / per));
    if (w % 16 == 15) printf("\n");
  }
  printf("\n");
  teardown(p);
}

// A kernel on ANOTHER stream holds `hogs` CUs (96 KB of LDS each: no GEMM workgroup fits beside it) while the
// GEMM runs -- the situation of a data-parallel step, where RCCL's all-reduce kernels share the chip with the
// backward GEMMs.  The persistent kernel's grid is sized for an empty chip, so some of its workgroups become
// resident only when others exit: with static work lists they then still own a full share of the tiles; with
// the dynamic hand-out they find what is left.
__global__ void hog_kernel(long long cycles, int* sink) {
  extern __shared__ int hog_lds[];
  const long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  int x = 0;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < cycles) { hog_lds[threadIdx.x] = x; x += hog_lds[(threadIdx.x + 1) & 63]; __builtin_amdgcn_s_sleep(8); }
  if (x == 0x7fffffff) *sink = x;
}
static void mode_contend(int first, int last, int hogs) {
  hipStream_t side;
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  int* sink;
  CK(hipMalloc(&sink, 4));
  printf("%d CUs held by another stream's kernel during the launch; TFLOP/s (and ms)\n", hogs);
  for (int si = first; si < last; si++) {
    Problem p;
    setup(p, SHAPES[si]);
    const size_t nc = (size_t)p.s.M * p.s.N;
    CK(hipMemset(p.C, 0, nc * 4));
    run(p, 1, p.C);
    CK(hipDeviceSynchronize());
    const float alone = time_launches(p, 13, p.C2, 4);
    double res[2];
    int okk = 1;
    for (int k = 0; k < 2; k++) {
      const int variant = k == 0 ? 14 : 13;       // static lists, whole tiles / dynamic hand-out + stream-K
      std::vector<double> t;
      for (int rep = 0; rep < 5; rep++) {
        if (p.s.sk > 1) CK(hipMemset(p.C2, 0, nc * 4));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(hog_kernel, dim3(hogs), dim3(64), 96 * 1024, side, (long long)(alone * 4.0e-3 * 100e6), sink);   // ~4x the launch, at 100 MHz
        usleep(200);                                // let the hogs take their CUs
        t.push_back(time_launches(p, variant, p.C2, 1));
        CK(hipDeviceSynchronize());
      }
      if (max_diff(p, 1) > 2e-5) okk = 0;
      std::sort(t.begin(), t.end());
      res[k] = t[2];
    }
    printf("%-9s %6d %5d %6d | alone %6.1f (%.3f ms) | static lists %6.1f (%.3f ms) | dynamic hand-out %6.1f (%.3f ms)%s\n", p.s.name,
           p.s.M, p.s.N, p.s.K, tflops(p.s, alone), alone, tflops(p.s, res[0]), res[0], tflops(p.s, res[1]), res[1], okk ? "" : "  WRONG");
    fflush(stdout);
    teardown(p);
  }
  CK(hipFree(sink));
}

// Each launch timed on its own (events on the stream), (a) back to back, (b) with a memory-bound kernel
// (a 1-GiB fill, ~0.25 ms) between the launches, (c) with a host synchronisation before every launch --
// the three situations a GEMM meets inside the model's step.
static void mode_interleave(int variant, int first, int last) {
  float* scratch;
  const size_t nscr = (size_t)256 << 20;
  CK(hipMalloc(&scratch, nscr * 4));
  for (int si = first; si < last; si++) {
    Problem p;
    setup(p, SHAPES[si]);
    const int n = 24;
    std::vector<hipEvent_t> e0(n), e1(n);
    for (int i = 0; i < n; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    time_launches(p, variant, p.C, 8);
    double med[5];
    for (int mode = 0; mode < 5; mode++) {
      for (int i = 0; i < n; i++) {
        if (mode == 1) hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, scratch, nscr, (unsigned)i);
        if (mode >= 2) CK(hipDeviceSynchronize());
        if (mode == 3) usleep(500);          // GPU idle for 0.5 ms / 3 ms before the launch
        if (mode == 4) usleep(3000);
        CK(hipEventRecord(e0[i], 0));
        run(p, variant, p.C);
        CK(hipEventRecord(e1[i], 0));
      }
      CK(hipDeviceSynchronize());
      std::vector<double> t;
      for (int i = 0; i < n; i++) { float ms; CK(hipEventElapsedTime(&ms, e0[i], e1[i])); t.push_back(tflops(p.s, ms)); }
      std::sort(t.begin(), t.end());
      med[mode] = t[n / 2];
    }
    printf("%-9s %6d %5d %6d | back-to-back %6.1f | after a 1-GiB fill %6.1f | after a host sync %6.1f | after 0.5 ms idle %6.1f | after 3 ms idle %6.1f  TFLOP/s (median of %d)\n",
           p.s.name, p.s.M, p.s.N, p.s.K, med[0], med[1], med[2], med[3], med[4], n);
    fflush(stdout);
    teardown(p);
  }
  CK(hipFree(scratch));
}

static void mode_probe(int first, int last) {
  for (int si = first; si < last; si++) {
    Problem p;
    setup(p, SHAPES[si]);
    const size_t nc = (size_t)p.s.M * p.s.N;
    for (int w = 0; w < 5; w++) run(p, 1, p.C);
    CK(hipMemset(p.probe, 0, 4096 * AIT_PROBE_WORDS * 8));
    if (p.s.sk > 1) CK(hipMemset(p.C, 0, nc * 4));
    for (int w = 0; w < 3; w++) run(p, 2, p.C);       // the last stamped launch is analysed
    CK(hipDeviceSynchronize());
    const float ms_plain = time_launches(p, 1, p.C, 10);
    const float ms_probe = time_launches(p, 2, p.C, 10);
    std::vector<unsigned long long> h(4096 * AIT_PROBE_WORDS);
    CK(hipMemcpy(h.data(), p.probe, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<int> ids;
    for (int b = 0; b < 4096; b++) if (h[(size_t)b * AIT_PROBE_WORDS + 1]) ids.push_back(b);
    if (ids.empty()) { printf("%s: no stamps\n", p.s.name); teardown(p); continue; }
    unsigned long long t0 = ~0ull, t1 = 0, first_end = ~0ull, last_start = 0;
    double loop = 0, wait = 0, bar = 0, slabs = 0, tiles = 0, clk_c = 0, clk_t = 0;
    std::vector<double> dur;
    std::map<unsigned long long, int> per_cu;
    for (int b : ids) {
      const unsigned long long* q = &h[(size_t)b * AIT_PROBE_WORDS];
      t0 = std::min(t0, q[0]); t1 = std::max(t1, q[1]);
      first_end = std::min(first_end, q[1]); last_start = std::max(last_start, q[0]);
      dur.push_back((q[1] - q[0]) * 0.01);
      loop += q[2]; wait += q[3]; bar += q[4]; slabs += q[5]; tiles += q[6];
      clk_c += (double)(q[9] - q[8]); clk_t += (double)(q[1] - q[0]) * 10e-9;
      const unsigned hw = (unsigned)q[7], xcc = (unsigned)(q[7] >> 32) & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    std::sort(dur.begin(), dur.end());
    const double span = (t1 - t0) * 0.01;
    // idle slot-time at the tail: sum over workgroups of (kernel end - own end), relative to slots x span
    double tail = 0, head = 0;
    for (int b : ids) {
      const unsigned long long* q = &h[(size_t)b * AIT_PROBE_WORDS];
      tail += (t1 - q[1]) * 0.01; head += (q[0] - t0) * 0.01;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    const int waves_per_simd = (p.s.ta && !p.s.tb) ? 2 : 4;    // resident waves sharing one matrix pipe
    const int mfma_per_slab = (p.s.ta && !p.s.tb) ? 64 : 32;
    printf("%-9s M=%d N=%d K=%d sk=%d: %d workgroups on %zu CUs (", p.s.name, p.s.M, p.s.N, p.s.K, p.s.sk, (int)ids.size(), per_cu.size());
    for (auto& kv : hist) printf("%d CUs x %d wg ", kv.second, kv.first);
    printf(")\n  plain %.1f us = %.1f TF/s; stamped %.1f us (+%.1f %%); span of stamped launch %.1f us\n", ms_plain * 1e3,
           tflops(p.s, ms_plain), ms_probe * 1e3, 100.0 * (ms_probe / ms_plain - 1), span);
    printf("  workgroup duration us: min %.1f med %.1f max %.1f;  first end at %.1f us, last start at %.1f us\n", dur.front(),
           dur[dur.size() / 2], dur.back(), (first_end - t0) * 0.01, (last_start - t0) * 0.01);
    printf("  idle slot-time: head %.2f %%, tail %.2f %% of slots x span\n", 100.0 * head / (ids.size() * span), 100.0 * tail / (ids.size() * span));
    printf("  wave 0: %.0f cycles per slab (matrix-pipe floor %d x 64 x %d waves/SIMD = %d); vmcnt wait %.2f %%, barrier %.2f %% of the slab loop; %.1f tiles and %.0f slabs per workgroup\n",
           loop / slabs, mfma_per_slab, waves_per_simd, mfma_per_slab * 64 * waves_per_simd, 100.0 * wait / loop, 100.0 * bar / loop, tiles / ids.size(), slabs / ids.size());
    printf("  slab loop = %.1f %% of workgroup lifetime (cycles); shader clock %.2f GHz\n", 100.0 * loop / clk_c, clk_c / clk_t * 1e-9);
    {  // per wave: share of its slab loop in the vmcnt wait / at the barrier (mean over workgroups)
      double wl[8] = {}, ww[8] = {}, wb[8] = {};
      for (int b : ids) {
        const unsigned long long* q = &h[(size_t)b * AIT_PROBE_WORDS];
        for (int w = 0; w < 8; w++) { wl[w] += q[10 + w * 3]; ww[w] += q[11 + w * 3]; wb[w] += q[12 + w * 3]; }
      }
      printf("  per wave (compute / vmcnt wait / barrier cycles per slab):");
      for (int w = 0; w < 8; w++) if (wl[w] > 0) printf("  w%d %.0f/%.0f/%.0f", w, (wl[w] - ww[w] - wb[w]) / slabs, ww[w] / slabs, wb[w] / slabs);
      printf("\n");
    }
    {  // the two workgroups of a CU: the one that ends first ("first") and its partner ("second")
      std::map<unsigned long long, std::vector<int>> by_cu;
      for (int b : ids) {
        const unsigned long long* q = &h[(size_t)b * AIT_PROBE_WORDS];
        const unsigned hw = (unsigned)q[7], xcc = (unsigned)(q[7] >> 32) & 0xf;
        by_cu[((unsigned long long)xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)].push_back(b);
      }
      double fa = 0, fb = 0, ea = 0, eb = 0, cu_end = 0; int n2 = 0; double min_cu_end = 1e30, max_cu_end = 0;
      for (auto& kv : by_cu) {
        double last = 0;
        for (int b : kv.second) last = std::max(last, (double)(h[(size_t)b * AIT_PROBE_WORDS + 1] - t0) * 0.01);
        min_cu_end = std::min(min_cu_end, last); max_cu_end = std::max(max_cu_end, last); cu_end += last;
        if (kv.second.size() != 2) continue;
        const unsigned long long* q0 = &h[(size_t)kv.second[0] * AIT_PROBE_WORDS];
        const unsigned long long* q1 = &h[(size_t)kv.second[1] * AIT_PROBE_WORDS];
        const unsigned long long* A = q0[1] <= q1[1] ? q0 : q1;
        const unsigned long long* B = q0[1] <= q1[1] ? q1 : q0;
        fa += (double)A[2] / A[5]; fb += (double)B[2] / B[5];
        ea += (A[1] - t0) * 0.01; eb += (B[1] - t0) * 0.01; n2++;
      }
      if (n2) printf("  per CU (2 workgroups): first ends at %.1f us with %.0f cycles/slab, second at %.1f us with %.0f cycles/slab\n",
                     ea / n2, fa / n2, eb / n2, fb / n2);
      printf("  CU end times: min %.1f mean %.1f max %.1f us (span %.1f)\n", min_cu_end, cu_end / by_cu.size(), max_cu_end, span);
    }
    fflush(stdout);
    teardown(p);
  }
}


// ---- micro: what one SIMD loses to the instructions around its MFMAs --------------------------------------
// Every wave runs `iters` blocks of 32 independent v_mfma_f32_32x32x2_f32 (4 accumulators) with, spread
// between them, G global_load_lds_dwordx4 (from an L2-resident buffer into its own LDS region), R
// ds_read_b128 (results consumed by the MFMAs of the next block) and optionally one barrier per block.
template <int G, int R, bool BAR>
__global__ __launch_bounds__(512, 2) void micro_kernel(const float* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float4 x[8];
  for (int i = 0; i < 8; i++) x[i] = make_float4(lane * 1e-3f + i, 1.f, 2.f, 3.f);
  float* mine = lds + wave * 2048;                      // 8 KB per wave
  const float* gp = src + ((size_t)blockIdx.x * 8 + wave) * 4096 + lane * 4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const float fa = x[j][a & 3], fb = x[(j + 1) & 7][(a + 1) & 3];
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (j < G) glds16(gp + (j & 3) * 256, mine + (j & 1) * 256);
      if (j < R) x[j] = *reinterpret_cast<const float4*>(mine + 512 + ((lane * 4 + j * 256) & 1023));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (G > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  float sacc = 0.f;
  for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) sacc += acc[a][r];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sacc;
}
template <int G, int R, bool BAR>
static void micro_case(const float* src, float* out, int wgs_per_cu) {
  const int iters = 4000, blocks = 256 * wgs_per_cu;
  auto k = micro_kernel<G, R, BAR>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 72 * 1024, 0, src, out, 200);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 72 * 1024, 0, src, out, iters);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = (double)blocks * 8 * iters * 32.0 * 4096.0;
  printf("  %d wg/CU  glds/32mfma %d  ds_read_b128/32mfma %d  barrier %d : %6.1f TF/s\n", wgs_per_cu, G, R, (int)BAR, fl / ms / 1e9);
  fflush(stdout);
}
static void mode_micro() {
  float *src, *out;
  CK(hipMalloc(&src, (size_t)512 * 8 * 4096 * 4 + 65536)); CK(hipMalloc(&out, (size_t)512 * 512 * 4));
  fill_kernel<<<1024, 256>>>(src, (size_t)512 * 8 * 4096, 0x42u);
  CK(hipDeviceSynchronize());
  for (int w = 1; w <= 2; w++) {
    micro_case<0, 0, false>(src, out, w);
    micro_case<0, 8, false>(src, out, w);
    micro_case<0, 8, true>(src, out, w);
    micro_case<1, 8, true>(src, out, w);
    micro_case<3, 8, true>(src, out, w);
    micro_case<3, 8, false>(src, out, w);
    micro_case<3, 0, false>(src, out, w);
    micro_case<6, 8, true>(src, out, w);
    micro_case<6, 0, false>(src, out, w);
  }
  hipFree(src); hipFree(out);
}

static void mode_pmc(int si, int variant, int n) {
  Problem p;
  setup(p, SHAPES[si]);
  for (int i = 0; i < n; i++) run(p, variant, p.C);
  CK(hipDeviceSynchronize());
  teardown(p);
}

// where a variant's result differs from the whole-tile reference: counts per 256 x 128 tile, first few elements
static void mode_diffmap(int si, int variant) {
  Problem p;
  setup(p, SHAPES[si]);
  const size_t nc = (size_t)p.s.M * p.s.N;
  CK(hipMemset(p.C, 0, nc * 4)); CK(hipMemset(p.C2, 0, nc * 4));
  run(p, getenv("LAB_SELF") ? variant : 1, p.C);      // LAB_SELF: the variant against a second launch of itself
  if (getenv("LAB_NOWS")) g_use_ws = false;
  run(p, variant, p.C2);
  g_use_ws = true;
  CK(hipDeviceSynchronize());
  std::vector<float> a(nc), b(nc);
  CK(hipMemcpy(a.data(), p.C, nc * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), p.C2, nc * 4, hipMemcpyDeviceToHost));
  double ref = 0;
  for (size_t i = 0; i < nc; i++) ref = std::max(ref, (double)fabsf(a[i]));
  const int tm = (p.s.M + 255) / 256, tn = (p.s.N + 127) / 128;
  std::vector<int> cnt((size_t)tm * tn, 0);
  int shown = 0;
  for (int r = 0; r < p.s.M; r++)
    for (int c = 0; c < p.s.N; c++) {
      const size_t i = (size_t)r * p.s.N + c;
      if (fabsf(a[i] - b[i]) > (getenv("LAB_SELF") ? 0.0 : 2e-5 * ref)) {
        cnt[(size_t)(r / 256) * tn + c / 128]++;
        if (shown++ < 12) printf("  (%d, %d) tile (%d, %d) in-tile (%d, %d): ref %.6g got %.6g\n", r, c, r / 256, c / 128, r % 256, c % 128, a[i], b[i]);
      }
    }
  {   // which accumulator: wave tile 128 x 64 -> (a, b) MFMA tile, register r, lane half lk, lane quarter
    int hist[4][2] = {}, regs[16] = {}, quarter[4] = {};
    for (int r = 0; r < p.s.M; r++)
      for (int c = 0; c < p.s.N; c++) {
        const size_t i = (size_t)r * p.s.N + c;
        if (fabsf(a[i] - b[i]) > (getenv("LAB_SELF") ? 0.0 : 2e-5 * ref)) {
          const int ir = r % 128, ic = c % 64, rr = ir % 32, cc = ic % 32;
          hist[ir / 32][ic / 32]++;
          const int lk = (rr >> 2) & 1, reg = (rr & 3) + 4 * (rr >> 3);
          regs[reg]++;
          quarter[lk * 2 + cc / 16]++;
        }
      }
    for (int x = 0; x < 4; x++) printf("a=%d: b=0 %d, b=1 %d\n", x, hist[x][0], hist[x][1]);
    printf("acc register:"); for (int x = 0; x < 16; x++) printf(" %d", regs[x]); printf("\nlane quarter:");
    for (int x = 0; x < 4; x++) printf(" %d", quarter[x]); printf("\n");
  }
  int bad = 0;
  for (int t = 0; t < tm * tn; t++)
    if (cnt[t]) { if (bad++ < 40) printf("tile %d (%d, %d): %d elements differ\n", t, t / tn, t % tn, cnt[t]); }
  printf("%d of %d tiles differ\n", bad, tm * tn);
  teardown(p);
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "ab";
  if (getenv("LAB_SLOTS")) g_slots = atoi(getenv("LAB_SLOTS"));
  int first = 0, last = NSHAPES;
  if (getenv("LAB_SHAPES")) { sscanf(getenv("LAB_SHAPES"), "%d:%d", &first, &last); }
  if (!strcmp(mode, "ab")) mode_ab(argc > 2 ? atoi(argv[2]) : 5, first, last);
  else if (!strcmp(mode, "probe")) mode_probe(first, last);
  else if (!strcmp(mode, "micro")) mode_micro();
  else if (!strcmp(mode, "diffmap")) mode_diffmap(atoi(argv[2]), atoi(argv[3]));
  else if (!strcmp(mode, "sweep")) mode_sweep(argc > 2 ? atoi(argv[2]) : 5, first, last);
  else if (!strcmp(mode, "contend")) mode_contend(first, last, argc > 2 ? atoi(argv[2]) : 64);
  else if (!strcmp(mode, "interleave")) mode_interleave(argc > 2 ? atoi(argv[2]) : 13, first, last);
  else if (!strcmp(mode, "sustain")) mode_sustain(argc > 2 ? atoi(argv[2]) : 13, argc > 3 ? atof(argv[3]) : 3.0, first);
  else if (!strcmp(mode, "pmc")) mode_pmc(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 10);
  else { printf("unknown mode\n"); return 1; }
  return 0;
}
