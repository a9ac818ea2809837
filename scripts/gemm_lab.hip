// scripts/gemm_lab.hip -- measuring harness for the fp32 MFMA GEMM (NOT part of the product library).
// Stand-alone executable (no Python on the GPU box):
//     scripts/build_gemm_lab.sh            (here: cross-compiles for gfx950)
//     scripts/_gemm_lab ab [rounds]        interleaved A/B of the round-1 kernel and the current one
//     scripts/_gemm_lab probe              one stamped launch per shape: workgroup time line, share of
//                                          the slab loop spent in the vmcnt wait / the barrier, clock
//     scripts/_gemm_lab pmc <shape> <variant> [n]   n launches of one variant (for rocprofv3 --pmc)
// The round-1 kernel comes from the history (81edb57), extracted by the build script into
// /tmp/ait_old_gemm_f32_impl.h with its namespace renamed; without it (-DNO_OLD) only the current one runs.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../ait_amd/csrc/gemm_f32_impl.h"
#ifndef NO_OLD
#include "/tmp/ait_old_gemm_f32_impl.h"
#endif

using namespace ait_gemm;

using NewD = Cfg<256, 128, 16, 4, 2, 4, MODE_DLDS>;
using NewD4 = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS>;
#ifndef NO_OLD
using OldD = ait_gemm_old::Cfg<256, 128, 16, 4, 2, 2, 6 + 256>;
using OldD4 = ait_gemm_old::Cfg<256, 128, 16, 2, 2, 2, 6 + 256>;
#endif

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
};
static const int NSHAPES = sizeof(SHAPES) / sizeof(SHAPES[0]);

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
  float *A, *B, *C, *C2, *bias;
  unsigned long long* probe;
  GemmArgs g;
};

static void setup(Problem& p, const Shape& s) {
  p.s = s;
  const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
  CK(hipMalloc(&p.A, na * 4)); CK(hipMalloc(&p.B, nb * 4)); CK(hipMalloc(&p.C, nc * 4)); CK(hipMalloc(&p.C2, nc * 4));
  CK(hipMalloc(&p.bias, (size_t)s.N * 4));
  CK(hipMalloc(&p.probe, 4096 * AIT_PROBE_WORDS * 8));
  fill_kernel<<<2048, 256>>>(p.A, na, 0x1234567u);
  fill_kernel<<<2048, 256>>>(p.B, nb, 0x89abcdeu);
  fill_kernel<<<64, 256>>>(p.bias, (size_t)s.N, 0x5555u);
  CK(hipMemset(p.C, 0, nc * 4)); CK(hipMemset(p.C2, 0, nc * 4));
  CK(hipDeviceSynchronize());
  const int lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
  const int flags = s.flags | (s.sk > 1 ? AIT_GEMM_ATOMIC : 0);
  int rc = make_args(s.ta, s.tb, s.M, s.N, s.K, 1.f, p.A, lda, p.B, ldb, p.C, s.N,
                     (s.flags & AIT_GEMM_RELU) ? p.bias : nullptr, nullptr, flags, s.sk, 0, 0, 16, p.g);
  if (rc) { printf("make_args rc %d\n", rc); exit(1); }
}
static void teardown(Problem& p) {
  hipFree(p.A); hipFree(p.B); hipFree(p.C); hipFree(p.C2); hipFree(p.bias); hipFree(p.probe);
}

// variant 0: round-1 kernel; 1: current kernel; 2: current kernel with the stamp probe
template <class Probe>
static int run_new(const GemmArgs& g, bool ak, bool bk, int slots) {
  const bool atomic = (g.flags & AIT_GEMM_ATOMIC) != 0;
  if (!ak && !bk) {
    return atomic ? launch<NewD4, false, false, EPI_ATOMIC, Probe>(g, 0, slots) : launch<NewD4, false, false, EPI_STORE, Probe>(g, 0, slots);
  }
  if (ak && bk) return atomic ? launch<NewD, true, true, EPI_ATOMIC, Probe>(g, 0, slots) : launch<NewD, true, true, EPI_STORE, Probe>(g, 0, slots);
  if (ak && !bk) return atomic ? launch<NewD, true, false, EPI_ATOMIC, Probe>(g, 0, slots) : launch<NewD, true, false, EPI_STORE, Probe>(g, 0, slots);
  return atomic ? launch<NewD, false, true, EPI_ATOMIC, Probe>(g, 0, slots) : launch<NewD, false, true, EPI_STORE, Probe>(g, 0, slots);
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
  if (variant == 1) return run_new<NoProbe>(g, ak, bk, g_slots);
  if (variant == 2) { g.probe = p.probe; return run_new<StampProbe>(g, ak, bk, g_slots); }
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
    fflush(stdout);
    teardown(p);
  }
}

static void mode_pmc(int si, int variant, int n) {
  Problem p;
  setup(p, SHAPES[si]);
  for (int i = 0; i < n; i++) run(p, variant, p.C);
  CK(hipDeviceSynchronize());
  teardown(p);
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "ab";
  if (getenv("LAB_SLOTS")) g_slots = atoi(getenv("LAB_SLOTS"));
  int first = 0, last = NSHAPES;
  if (getenv("LAB_SHAPES")) { sscanf(getenv("LAB_SHAPES"), "%d:%d", &first, &last); }
  if (!strcmp(mode, "ab")) mode_ab(argc > 2 ? atoi(argv[2]) : 5, first, last);
  else if (!strcmp(mode, "probe")) mode_probe(first, last);
  else if (!strcmp(mode, "pmc")) mode_pmc(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 10);
  else { printf("unknown mode\n"); return 1; }
  return 0;
}
