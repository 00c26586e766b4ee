// Stand-alone probe (not part of the product library): the K loop of the fp32 MFMA implicit-GEMM kernels with the
// operand tiles staged by LDS-DMA (global_load_lds_dwordx4) into an XOR-swizzled, unpadded LDS ring instead of
// global -> registers -> ds_write, and the per-K-tile barrier placed in the shadow of the tile's last MFMAs.
// Question it answers: how busy does ONE wave per SIMD keep the matrix pipe (1 workgroup per CU) compared with two
// (2 workgroups per CU), for the register-staged loop of conv_mfma.hip and for the LDS-DMA loop -- i.e. can the
// epilogue of one workgroup be hidden behind the K loop of the other.
//   C[M x N] = A[M x K] * B[N x K]^T, fp32, 128 x 128 tiles, BK = 32, v_mfma_f32_32x32x2_f32, natural-K order.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/glds_probe.hip -o gpurun_out/glds_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = BK * 4;               // bytes per LDS row (no padding)
constexpr int STAGE = (BM + BN) * ROWB;    // 32 KB
#define SB() __builtin_amdgcn_sched_barrier(0)

struct Args {
  const float *a, *b;
  float *c;
  int M, N, K;        // C is M x N; A rows alias modulo a_rows
  int a_rows;
  int epi;            // emulated epilogue: dependent sqrt/div rounds per accumulator element
  int stagger;        // s_sleep(127) units per HW wave slot for first-round workgroups
  int first_round;
  unsigned long long *clk;  // [2] per launch: sum over workgroups of shader cycles, of 100 MHz ticks
};

// LDS-DMA of 16 bytes per lane: LDS destination = lds_dst (wave-uniform) + lane * 16, source = base + voff (per lane)
__device__ __forceinline__ void glds16(const float *base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

template <int NS, int EXTRA_LDS>
__global__ __launch_bounds__(256, 2) void gemm_glds(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned long long t_c0 = clock64(), t_r0 = wall_clock64();
  if (g.stagger > 0 && blockIdx.x < (unsigned)g.first_round) {
    const uint32_t slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));  // HW_ID.WAVE_ID
    for (uint32_t i = 0; i < (slot & 1u) * (uint32_t)g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const int m0 = blockIdx.x * BM;
  const int K = g.K, nkt = K / BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // ---- loader: 4 A + 4 B instructions per wave and K-tile; instruction j of wave w covers rows 32 j + 8 w + (lane >> 3),
  // lane & 7 is the 16-byte slot; slot s of row R holds data chunk s ^ swz(R), swz(R) = (R & 7) ^ ((R >> 3) & 3)
  const int l3 = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ l3 ^ wave;  // = slot ^ swz(32 j + 8 w + l3) for every j
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 32 * j + 8 * wave + l3;
    a_off[j] = (uint32_t)(((m0 + r) % g.a_rows) * K + chunk * 4) * 4u;
    b_off[j] = (uint32_t)(r * K + chunk * 4) * 4u;
  }
  auto issue_tile = [&](int kt, int buf) {
    const uint32_t kb = (uint32_t)kt * (BK * 4);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + wave * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(g.a, a_off[j] + kb, dst + j * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(g.b, b_off[j] + kb, dst + BM * ROWB + j * 4096);
  };

  // ---- fragment addressing: lane reads row (lane & 31) of its 32-row block, data chunk 2 o + (lane >> 5)
  const int sw = (lane & 7) ^ ((lane >> 3) & 3), hh = lane >> 5;
  uint32_t foff[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) foff[o] = (uint32_t)(((2 * o + hh) ^ sw) << 4);
  const char *a_base = smem + (wm * 64 + (lane & 31)) * ROWB;
  const char *b_base = smem + BM * ROWB + (wn * 64 + (lane & 31)) * ROWB;

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f4 fa[2][2], fb[2][2];  // [set][block]
  auto read_oct = [&](int set, int buf, int o) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const f4 *>(a_base + buf * STAGE + i * 32 * ROWB + foff[o]);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const f4 *>(b_base + buf * STAGE + j * 32 * ROWB + foff[o]);
  };
  auto mfma_step = [&](int set, int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][s], fb[set][j][s], acc[i][j], 0, 0, 0);
  };

  issue_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nkt > 1) issue_tile(1, 1 % NS);
  if (NS > 2 && nkt > 2) issue_tile(2, 2);
  read_oct(0, 0, 0);
  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int nbuf = buf + 1 == NS ? 0 : buf + 1;
    // octets 0..2: reads of the next octet go out first, then this octet's 16 MFMAs
    read_oct(1, buf, 1);
    SB();
    mfma_step(0, 0); mfma_step(0, 1); mfma_step(0, 2); mfma_step(0, 3);
    SB();
    read_oct(0, buf, 2);
    SB();
    mfma_step(1, 0); mfma_step(1, 1); mfma_step(1, 2); mfma_step(1, 3);
    SB();
    read_oct(1, buf, 3);
    SB();
    mfma_step(0, 0); mfma_step(0, 1); mfma_step(0, 2); mfma_step(0, 3);
    SB();
    // octet 3: in the shadow of its MFMAs -- next tile landed (own DMAs) + barrier (everybody's; and everybody has
    // read this buffer: the octet-3 fragments are in registers), first octet of the next tile, refill of this buffer
    mfma_step(1, 0);
    SB();
    if (kt + 1 < nkt) {
      if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the newest tile (8 DMAs) may still be in flight
      __builtin_amdgcn_s_barrier();
    }
    SB();
    mfma_step(1, 1);
    SB();
    if (kt + 1 < nkt) read_oct(0, nbuf, 0);
    SB();
    mfma_step(1, 2);
    SB();
    if (kt + NS < nkt) issue_tile(kt + NS, buf);
    else if (NS > 2) asm volatile("s_nop 0" ::: "memory");
    SB();
    mfma_step(1, 3);
    SB();
    buf = nbuf;
  }
  if (NS > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- emulated epilogue: dependent IEEE sqrt + division rounds per element (VALU only, matrix pipe idle)
  for (int e = 0; e < g.epi; ++e) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          acc[i][j][r] = v / __builtin_sqrtf(v * v + 1.0f);
        }
  }
  const int col = wn * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
      for (int j = 0; j < 2; ++j) g.c[(size_t)row * g.N + col + 32 * j] = acc[i][j][r];
    }
  if (tid == 0) {
    atomicAdd(g.clk, clock64() - t_c0);
    atomicAdd(g.clk + 1, wall_clock64() - t_r0);
  }
}

// ---- the register-staged loop of conv_mfma.hip (FASTK conv, 128 x 128): one LDS buffer, two barriers per K-tile
template <int EXTRA>
__global__ __launch_bounds__(256, 2) void gemm_regs(Args g) {
  constexpr int STRIDE = BK + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  float *smem = reinterpret_cast<float *>(smem_);
  float *As = smem, *Bs = smem + BM * STRIDE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned long long t_c0 = clock64(), t_r0 = wall_clock64();
  if (g.stagger > 0 && blockIdx.x < (unsigned)g.first_round) {
    const uint32_t slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));
    for (uint32_t i = 0; i < (slot & 1u) * (uint32_t)g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const int m0 = blockIdx.x * BM;
  const int K = g.K, nkt = K / BK;
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = tid + 256 * j, r = u >> 2, oct = u & 3;
    a_off[j] = (uint32_t)(((m0 + r) % g.a_rows) * K + oct * 8);
    b_off[j] = (uint32_t)(r * K + oct * 8);
  }
  f4 ra[2][2], rb[2][2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float *pa = g.a + a_off[j] + kt * BK, *pb = g.b + b_off[j] + kt * BK;
      ra[j][0] = *reinterpret_cast<const f4 *>(pa);
      ra[j][1] = *reinterpret_cast<const f4 *>(pa + 4);
      rb[j][0] = *reinterpret_cast<const f4 *>(pb);
      rb[j][1] = *reinterpret_cast<const f4 *>(pb + 4);
    }
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float *a_frag = As + (wm * 64 + (lane & 31)) * STRIDE + (lane >> 5) * 4;
  const float *b_frag = Bs + (wn * 64 + (lane & 31)) * STRIDE + (lane >> 5) * 4;
  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int u = tid + 256 * j;
      float *da = As + (u >> 2) * STRIDE + (u & 3) * 8, *db = Bs + (u >> 2) * STRIDE + (u & 3) * 8;
      *reinterpret_cast<f4 *>(da) = ra[j][0];
      *reinterpret_cast<f4 *>(da + 4) = ra[j][1];
      *reinterpret_cast<f4 *>(db) = rb[j][0];
      *reinterpret_cast<f4 *>(db + 4) = rb[j][1];
    }
    __syncthreads();
    if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f4 *>(a_frag + i * 32 * STRIDE + o * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f4 *>(b_frag + j * 32 * STRIDE + o * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  }
  for (int e = 0; e < g.epi; ++e) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          acc[i][j][r] = v / __builtin_sqrtf(v * v + 1.0f);
        }
  }
  const int col = wn * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
      for (int j = 0; j < 2; ++j) g.c[(size_t)row * g.N + col + 32 * j] = acc[i][j][r];
    }
  if (tid == 0) {
    atomicAdd(g.clk, clock64() - t_c0);
    atomicAdd(g.clk + 1, wall_clock64() - t_r0);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// reference in the order of the arithmetic contract: groups of 8 in the order 0,4,1,5,2,6,3,7, fmaf chain from +0
static float ref_dot(const float *a, const float *b, int K) {
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 8)
    for (int s = 0; s < 4; ++s) {
      acc = fmaf(a[k0 + s], b[k0 + s], acc);
      acc = fmaf(a[k0 + 4 + s], b[k0 + 4 + s], acc);
    }
  return acc;
}

int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 1600;
  const int tiles = argc > 2 ? atoi(argv[2]) : 16200;
  const int a_rows = 65536, N = 128;
  const int M = tiles * BM;
  std::vector<float> ha((size_t)a_rows * K), hb((size_t)N * K);
  srand(1);
  for (auto &v : ha) v = (float)(rand() % 2001 - 1000) / 1000.f;
  for (auto &v : hb) v = (float)(rand() % 2001 - 1000) / 1000.f;
  float *da, *db, *dc;
  CK(hipMalloc(&da, ha.size() * 4));
  CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  unsigned long long *dclk;
  CK(hipMalloc(&dclk, 16));
  Args g{da, db, dc, M, N, K, a_rows, 0, 0, 512, dclk};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> hc((size_t)1024 * N);
  auto check = [&](const char *name) {
    CK(hipMemcpy(hc.data(), dc + (size_t)(M - 1024) * N, hc.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < 1024; r += 37)
      for (int c = 0; c < N; c += 5) {
        const float want = ref_dot(&ha[(size_t)((M - 1024 + r) % a_rows) * K], &hb[(size_t)c * K], K);
        if (memcmp(&want, &hc[(size_t)r * N + c], 4) != 0) ++bad;
      }
    printf("  %-28s check: %s\n", name, bad ? "MISMATCH" : "bit-exact");
  };
  auto run = [&](const char *name, auto kernel, size_t lds, int epi, int stagger) {
    g.epi = epi;
    g.stagger = stagger;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), lds, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipMemset(dclk, 0, 16));
    CK(hipEventRecord(e0));
    const int n = 5;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), lds, 0, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
    unsigned long long hclk[2];
    CK(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
    const double tf = 2.0 * M * N * K / ms / 1e9;
    printf("%-34s lds %6zu epi %2d stagger %3d: %8.3f ms  %6.1f TFLOP/s  clk %.3f GHz -> %.3f of the peak at that clock, WG avg %.0f kcyc\n", name, lds,
           epi, stagger, ms, tf, ghz, tf / (256 * 4 * 64 * 2 * ghz * 1e-3), (double)hclk[0] / (5.0 * tiles) / 1e3);
  };
  const size_t one = 96 * 1024;  // forces one workgroup per CU
  for (int epi : {0, 2}) {
    run("regs  2 WG/CU", gemm_regs<0>, (BM + BN) * (BK + 4) * 4, epi, 0);
    if (!epi) check("regs");
    run("regs  1 WG/CU", gemm_regs<0>, one, epi, 0);
    run("glds2 2 WG/CU", gemm_glds<2, 0>, 2 * STAGE, epi, 0);
    if (!epi) check("glds2");
    run("glds2 1 WG/CU", gemm_glds<2, 0>, one, epi, 0);
    run("glds3 1 WG/CU", gemm_glds<3, 0>, 3 * STAGE, epi, 0);
    if (!epi) check("glds3");
    if (epi) {
      for (int st : {8, 16, 32, 64}) {
        run("regs  2 WG/CU staggered", gemm_regs<0>, (BM + BN) * (BK + 4) * 4, epi, st);
        run("glds2 2 WG/CU staggered", gemm_glds<2, 0>, 2 * STAGE, epi, st);
      }
    }
  }
  return 0;
}
