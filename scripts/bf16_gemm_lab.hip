// scripts/bf16_gemm_lab.hip -- LAB ONLY (not part of libait_hip.so): what a GEMM with bf16 operands STORED in memory
// reaches on MI355X with the product kernel's structure (LDS-DMA slabs, XOR-swizzled row images, 128x64 per wave).
// VERDICT round 2 item 8 asks for "bf16 GEMM >= 1000 TFLOP/s in the lab"; the product's bf16 mode keeps f32 tensors in
// memory (operands rounded in registers) and is bound by that traffic at 345-470 TFLOP/s.
//   C[M,N] (f32) = A[M,K] (bf16, K contiguous) . B[N,K]^T (bf16, K contiguous); M, N multiples of 256, K of 64.
//   tile 256 x 256 x 64, eight waves of 128 x 64 (two per SIMD), two 64-KB stages of LDS, one workgroup per CU;
//   global -> LDS with global_load_lds_dwordx4 (1 KB = 8 rows of 128 B per instruction), 16-B chunk c of row r stored
//   at chunk position c ^ (r & 7): the ds_read_b128 that fetches a lane's eight k-values is conflict-free.
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 scripts/bf16_gemm_lab.hip -o scripts/_bf16_gemm_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int STAGE = (BM + BN) * BK * 2;     // bytes per stage (64 KB)

__device__ __forceinline__ void glds16(const void* src, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "m0", "memory");
}

__global__ __launch_bounds__(NT, 2) void gemm_bf16_nt(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // PERSISTENT: one workgroup per CU walks a list of tiles with ONE continuous stream of slabs -- while the last slabs of
  // a tile are multiplied the first two of the next are already in flight, and the epilogue's stores drain under the
  // next tile's MFMAs.  XCD-aware order: blocks b and b + 8 share an XCD; every XCD owns a contiguous chunk of the tile
  // list (row-major over N inside an M panel: the A panel stays in one L2), its workgroups take every (G/8)-th tile of it.
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  const int per = (tiles + 7) / 8, wg_per_xcd = gridDim.x / 8;
  const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
  const int chunk_end = min(per, tiles - xcd * per);          // tiles in this XCD's chunk
  const int mine = j < chunk_end ? (chunk_end - j + wg_per_xcd - 1) / wg_per_xcd : 0;
  if (mine == 0) return;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int slabs = K / BK, total = mine * slabs;
  const int rr = lane >> 3, pos = lane & 7;                       // row within the granule, chunk position in LDS
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
  auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int t = xcd * per + j + i * wg_per_xcd;
    m0 = (t / tiles_n) * BM;
    n0 = (t % tiles_n) * BN;
  };
  // loader: per slab A is 32 granules of 1 KB (8 rows), B likewise; wave w takes granules w, w + 8, ...
  auto issue = [&](int g, int stage) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(g / slabs, m0, n0);
    const int k0 = (g % slabs) * BK;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int q = wave + i * 8;                                  // granule 0..31
      const int row = q * 8 + rr;
      const int c = pos ^ (row & 7);
      const unsigned short* sa = A + (size_t)(m0 + row) * K + k0 + c * 8;
      const unsigned da = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + q * 1024);
      glds16(sa, da);
      const unsigned short* sb = B + (size_t)(n0 + row) * K + k0 + c * 8;
      const unsigned db = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + BM * BK * 2 + q * 1024);
      glds16(sb, db);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  issue(0, 0);
  if (total > 1) issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int s = 0, ti = 0;
  for (int g = 0; g < total; g++) {
    const unsigned char* sa = lds + (g & 1) * STAGE;
    const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int c = 2 * ks + lk;
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int row = wm + a * 32 + li;
        fa[a] = *reinterpret_cast<const bf16x8*>(sa + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int row = wn + b * 32 + li;
        fb[b] = *reinterpret_cast<const bf16x8*>(sb + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    // slab g + 1 (requested one iteration ago) must have landed; then everyone is done with stage g & 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + 2 < total) issue(g + 2, g & 1);
    if (++s == slabs) {
      // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
      int m0, n0;
      tile_origin(ti, m0, n0);
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            C[(size_t)row * N + n0 + wn + b * 32 + li] = acc[a][b][r];
            acc[a][b][r] = 0.f;
          }
      s = 0;
      ti++;
    }
  }
}

static unsigned short to_bf16(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float from_bf16(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; } shapes[] = {{512, 512, 256}, {76800, 1536, 512}, {76800, 2048, 512}, {76800, 512, 2048},
                                            {76800, 1536, 4096}, {8192, 8192, 8192}};
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_nt), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  for (const Shape& sh : shapes) {
    const size_t na = (size_t)sh.M * sh.K, nb = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
    std::vector<unsigned short> ha(na), hb(nb);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = to_bf16(rnd());
    for (auto& v : hb) v = to_bf16(rnd());
    unsigned short *da, *db;
    float* dc;
    hipMalloc(&da, na * 2); hipMalloc(&db, nb * 2); hipMalloc(&dc, nc * 4);
    hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), nb * 2, hipMemcpyHostToDevice);
    const int tiles = (sh.M / BM) * (sh.N / BN);
    const int grid = tiles < 256 ? (tiles + 7) / 8 * 8 : 256;         // persistent: one workgroup per CU
    auto launch = [&]() { hipLaunchKernelGGL(gemm_bf16_nt, dim3(grid), dim3(NT), 2 * STAGE, 0, da, db, dc, sh.M, sh.N, sh.K); };
    launch();
    hipDeviceSynchronize();
    // spot check against a double-precision product of the same bf16 values
    std::vector<float> hc(nc);
    hipMemcpy(hc.data(), dc, nc * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int probe = 0; probe < 4000; probe++) {
      const int i = (int)((probe * 2654435761u) % (unsigned)sh.M), j = (int)((probe * 40503u + 17u) % (unsigned)sh.N);
      double ref = 0, mag = 0;
      for (int k = 0; k < sh.K; k++) {
        const double p = (double)from_bf16(ha[(size_t)i * sh.K + k]) * from_bf16(hb[(size_t)j * sh.K + k]);
        ref += p; mag += fabs(p);
      }
      worst = fmax(worst, fabs(hc[(size_t)i * sh.N + j] - ref) / (mag + 1e-30));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("M=%6d N=%5d K=%5d: %8.1f us  %7.1f TFLOP/s   max |err| / sum|a||b| = %.2e (4000 elements)\n", sh.M, sh.N, sh.K, ms * 1e3,
           2.0 * sh.M * sh.N * sh.K / ms / 1e9, worst);
    hipFree(da); hipFree(db); hipFree(dc);
  }
  return 0;
}
