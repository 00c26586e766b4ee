// Tuning aid (round 5): what ONE wavefront pays per instruction in the dependent chains the range coder is made of --
// SALU / VALU dependent issue, VALU <-> SALU crossings (v_readlane, vcc -> s_bcnt1), taken / not-taken branches, the
// 64-bit multiply-add, ds_bpermute.  One workgroup of 64 threads, each test = ITER iterations of an unrolled chain of
// 16 (or fewer) copies of the pattern, timed with s_memtime (shader clock).  Prints cycles per pattern copy.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
constexpr int ITER = 20000;

#define TEST_BEGIN(name)                                                                      \
  __global__ __launch_bounds__(64) void name(unsigned long long *out, unsigned *sink, unsigned seed) { \
    unsigned v0 = threadIdx.x + seed, v1 = seed * 3 + 1, v2 = 7, v3 = 0;                      \
    unsigned s0 = seed, s1 = seed & 31, s2 = 1, s3 = 0;                                        \
    unsigned long long w = ((unsigned long long)seed << 32) | 5;                               \
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));                                \
    asm volatile("" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");                                \
    const unsigned long long c0 = clock64();                                                   \
    for (int it = 0; it < ITER; ++it) {
#define TEST_END                                                                              \
    }                                                                                          \
    const unsigned long long c1 = clock64();                                                   \
    sink[threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + (unsigned)w;                   \
    if (threadIdx.x == 0) out[0] = c1 - c0;                                                    \
  }

TEST_BEGIN(t_empty)
  asm volatile("s_nop 0");
TEST_END

TEST_BEGIN(t_salu_add_dep)
  asm volatile(R16("s_add_u32 %0, %0, 1\n\t") : "+s"(s0) : : "scc");
TEST_END

TEST_BEGIN(t_salu_add_indep)
  asm volatile(R4("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t")
               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
TEST_END

TEST_BEGIN(t_salu_lshl64_dep)
  asm volatile(R16("s_lshl_b64 %0, %0, 1\n\t") : "+s"(w) : : "scc");
TEST_END

TEST_BEGIN(t_salu_flbit_dep)
  asm volatile(R16("s_flbit_i32_b32 %0, %0\n\t") : "+s"(s0) : : "scc");
TEST_END

TEST_BEGIN(t_salu_mul_dep)
  asm volatile(R16("s_mul_i32 %0, %0, %1\n\t") : "+s"(s0) : "s"(s2));
TEST_END

TEST_BEGIN(t_valu_add_dep)
  asm volatile(R16("v_add_u32 %0, %0, %1\n\t") : "+v"(v0) : "v"(v1));
TEST_END

TEST_BEGIN(t_valu_add_indep)
  asm volatile(R4("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t")
               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v1));
TEST_END

TEST_BEGIN(t_valu_ffbh_dep)
  asm volatile(R16("v_ffbh_u32 %0, %0\n\t") : "+v"(v0));
TEST_END

TEST_BEGIN(t_valu_bfe_dep)
  asm volatile(R16("v_bfe_u32 %0, %0, %1, %2\n\t") : "+v"(v0) : "v"(v1), "v"(v2));
TEST_END

TEST_BEGIN(t_valu_lshl64_dep)
  {
    unsigned long long vw = w + threadIdx.x;
    asm volatile(R16("v_lshlrev_b64 %0, %1, %0\n\t") : "+v"(vw) : "v"(v2));
    w = vw;
    asm volatile("" : "+s"(s3));
  }
TEST_END

TEST_BEGIN(t_valu_mad64_dep)
  {
    unsigned long long vw = w + threadIdx.x;
    // D = S0 * S1 + S2 (64-bit); the low half feeds the next multiply
    asm volatile(R16("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t") : "+v"(vw) : "v"(v1), "v"(v2) : "vcc");
    v3 += (unsigned)vw;
  }
TEST_END

TEST_BEGIN(t_valu_mad64_chain)  // result (>> 16) -> multiplicand of the next (the coder's hl -> t -> low -> hl chain)
  {
    unsigned long long vw = w + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(vw) : "v"(v1), "v"(v2) : "vcc");
      v1 = (unsigned)(vw >> 16);
      asm volatile("" : "+v"(v1));
    }
    v3 += (unsigned)vw;
  }
TEST_END

TEST_BEGIN(t_mad_u24_pair_dep)
  asm volatile(R16("v_mad_u32_u24 %0, %0, %1, %2\n\t") : "+v"(v0) : "v"(v1), "v"(v2));
TEST_END

TEST_BEGIN(t_readlane_roundtrip)  // VALU -> SGPR (readlane, lane from SGPR) -> SALU -> VALU
  asm volatile(R16("v_readlane_b32 %1, %0, %2\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v0), "+s"(s0) : "s"(s1) : "scc");
TEST_END

TEST_BEGIN(t_readlane_to_valu)  // VALU -> readlane -> VALU (SGPR operand), no SALU in between
  asm volatile(R16("v_readlane_b32 %1, %0, %2\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v0), "+s"(s0) : "s"(s1));
TEST_END

TEST_BEGIN(t_readfirstlane_roundtrip)
  asm volatile(R16("v_readfirstlane_b32 %1, %0\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v0), "+s"(s0) : : "scc");
TEST_END

TEST_BEGIN(t_cmp_bcnt_valu)  // v_cmp -> vcc -> s_bcnt1 -> VALU operand
  asm volatile(R16("v_cmp_le_u32 vcc, %0, %2\n\ts_bcnt1_i32_b64 %1, vcc\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v0), "+s"(s0) : "v"(v1) : "vcc", "scc");
TEST_END

TEST_BEGIN(t_cmp_bcnt_readlane_valu)  // the decoder's search: cmp -> popcount -> readlane at that lane -> VALU
  asm volatile(R16("v_cmp_le_u32 vcc, %0, %3\n\ts_bcnt1_i32_b64 %1, vcc\n\ts_and_b32 %1, %1, 63\n\tv_readlane_b32 %2, %0, %1\n\tv_add_u32 %0, %2, %0\n\t")
               : "+v"(v0), "+s"(s0), "+s"(s2) : "v"(v1) : "vcc", "scc");
TEST_END

TEST_BEGIN(t_cmp_vcc_cndmask)  // VALU compare feeding a VALU select through vcc
  asm volatile(R16("v_cmp_le_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t") : "+v"(v0) : "v"(v1), "v"(v2) : "vcc");
TEST_END

TEST_BEGIN(t_branch_not_taken)
  asm volatile(R16("s_cmp_eq_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n\t1:\n\t") : "+s"(s0) : : "scc");
TEST_END

TEST_BEGIN(t_branch_taken)
  asm volatile(R16("s_cmp_lg_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n\t1:\n\ts_add_u32 %0, %0, 2\n\t") : "+s"(s0) : : "scc");
TEST_END

TEST_BEGIN(t_branch_vcc_from_valu)  // v_cmp -> vcc -> s_cbranch_vccz (not taken)
  asm volatile(R16("v_cmp_eq_u32 vcc, %0, %1\n\ts_cbranch_vccnz 1f\n\tv_add_u32 %0, %0, 1\n\t1:\n\t") : "+v"(v0) : "v"(v3) : "vcc");
TEST_END

TEST_BEGIN(t_bpermute_dep)
  asm volatile(R16("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v0) : "v"(v2));
TEST_END

TEST_BEGIN(t_writelane)
  asm volatile(R16("v_writelane_b32 %0, %1, 3\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(v0), "+s"(s0) : : "scc");
TEST_END

TEST_BEGIN(t_salu_valu_alternate_indep)  // does independent SALU work issue under VALU work of the same wave?
  asm volatile(R16("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(v0), "+s"(s0) : "v"(v1) : "scc");
TEST_END

TEST_BEGIN(t_valu_dpp_rowshr_dep)
  asm volatile(R16("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(v0));
TEST_END

struct T {
  const char *name;
  void (*fn)(unsigned long long *, unsigned *, unsigned);
  int copies;
  const char *what;
};

int main() {
  unsigned long long *out;
  unsigned *sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 64 * 4);
  T tests[] = {
      {"empty loop (s_nop)", t_empty, 1, "loop overhead per iteration"},
      {"s_add_u32 dependent", t_salu_add_dep, 16, ""},
      {"s_add_u32 independent x4", t_salu_add_indep, 16, ""},
      {"s_lshl_b64 dependent", t_salu_lshl64_dep, 16, ""},
      {"s_flbit_i32_b32 dependent", t_salu_flbit_dep, 16, ""},
      {"s_mul_i32 dependent", t_salu_mul_dep, 16, ""},
      {"v_add_u32 dependent", t_valu_add_dep, 16, ""},
      {"v_add_u32 independent x4", t_valu_add_indep, 16, ""},
      {"v_ffbh_u32 dependent", t_valu_ffbh_dep, 16, ""},
      {"v_bfe_u32 dependent", t_valu_bfe_dep, 16, ""},
      {"v_lshlrev_b64 dependent", t_valu_lshl64_dep, 16, ""},
      {"v_mad_u64_u32 dependent (addend)", t_valu_mad64_dep, 16, ""},
      {"v_mad_u64_u32 + v_alignbit chain", t_valu_mad64_chain, 16, "per (mad, alignbit) pair"},
      {"v_mad_u32_u24 dependent", t_mad_u24_pair_dep, 16, ""},
      {"v_readlane -> s_add -> v_add", t_readlane_roundtrip, 16, "per round trip (3 instr)"},
      {"v_readlane -> v_add", t_readlane_to_valu, 16, "per pair"},
      {"v_readfirstlane -> s_add -> v_add", t_readfirstlane_roundtrip, 16, "per round trip"},
      {"v_cmp -> s_bcnt1 -> v_add", t_cmp_bcnt_valu, 16, "per triple"},
      {"v_cmp -> s_bcnt1 -> s_and -> v_readlane -> v_add", t_cmp_bcnt_readlane_valu, 16, "per 5 instr"},
      {"v_cmp -> v_cndmask (vcc)", t_cmp_vcc_cndmask, 16, "per pair"},
      {"s_cmp + s_cbranch not taken + s_add", t_branch_not_taken, 16, "per 3 instr"},
      {"s_cmp + s_cbranch TAKEN + s_add", t_branch_taken, 16, "per 3 instr executed"},
      {"v_cmp + s_cbranch_vccnz not taken + v_add", t_branch_vcc_from_valu, 16, "per 3 instr"},
      {"ds_bpermute_b32 + wait dependent", t_bpermute_dep, 16, ""},
      {"v_writelane + s_add", t_writelane, 16, "per pair"},
      {"v_add + s_add alternating (independent)", t_salu_valu_alternate_indep, 16, "per pair"},
      {"v_mov_dpp row_shr dependent", t_valu_dpp_rowshr_dep, 16, ""},
  };
  double base = 0;
  for (auto &t : tests) {
    fprintf(stderr, "running %s\n", t.name);
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, out, sink, 12345u + rep);
      unsigned long long c = 0;
      hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    const double per_iter = (double)best / ITER;
    if (t.copies == 1) base = per_iter;
    fflush(stdout);
    printf("%-52s %8.2f cycles per copy   (%.1f per iteration, loop overhead %.1f) %s\n", t.name, (per_iter - (t.copies == 1 ? 0 : base)) / t.copies,
           per_iter, base, t.what);
  }
  return 0;
}
