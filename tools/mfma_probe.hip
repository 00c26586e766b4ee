// Stand-alone probe (not part of the product library):
//  1. sustained fp32 MFMA rate of this box with no memory traffic (the practical ceiling the
//     conv kernels are priced against next to the 157.3 TFLOP/s data-sheet peak);
//  2. whether v_mfma_f32_16x16x4_f32 accumulates k = 0..3 as one ordered fmaf chain, the property
//     conv_mfma.hip relies on for v_mfma_f32_32x32x2_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256) void peak32(float* out, int iters, float a, float b) {
    f16v acc[K];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Same loop with data-dependent operands (random mantissas, rotated every step) so the datapath toggles as
// it does in a real convolution: separates "issue-limited" from "power/clock-limited".
__global__ __launch_bounds__(256) void peak32_rand(float* out, const float* in, int iters) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(threadIdx.x * 8 + i) & 4095];
        b[i] = in[(threadIdx.x * 8 + i + 2048) & 4095];
    }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 + i], b[4 + i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The conv kernel's K loop rebuilt feature by feature (FEAT bits): operands always come from LDS with the
// kernel's row stride / fragment addressing (ds_read_b128 per 4 MFMA steps, 2x2 accumulator blocks per wave);
//   1 = the two barriers per K-tile, 2 = the 8 ds_write_b128 of the staged tile, 4 = the 8 global_load_dwordx4
// of the next tile (128-byte rows out of a large buffer), 8 = the even/odd K permutation (v_mov) before the writes.
typedef float f4 __attribute__((ext_vector_type(4)));
template <int FEAT>
__global__ __launch_bounds__(256, 2) void peak32_lds(float* out, const float* in, int iters, const float* big, unsigned big_mask) {
    constexpr int STRIDE = 36;
    __shared__ __attribute__((aligned(16))) float smem[256 * STRIDE];
    for (int i = threadIdx.x; i < 256 * STRIDE; i += 256) smem[i] = in[i & 4095];
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* a_frag = smem + (wm * 64 + (lane & 31)) * STRIDE + (lane >> 5) * 4;
    const float* b_frag = smem + 128 * STRIDE + (wn * 64 + (lane & 31)) * STRIDE + (lane >> 5) * 4;
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f4 st[4][2];
    for (int u = 0; u < 4; ++u)
        for (int q = 0; q < 2; ++q) st[u][q] = *reinterpret_cast<const f4*>(in + ((tid * 8 + u * 64 + q * 4) & 4095));
    for (int it = 0; it < iters; ++it) {
        if (FEAT & 1) __syncthreads();
        if (FEAT & 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float* dst = smem + ((tid + 256 * u) >> 2) * STRIDE + ((tid + 256 * u) & 3) * 8;
                if (FEAT & 32) {
                    // same permutation with v_pk_mov_b32: two dwords per instruction
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a01 = {st[u][0][0], st[u][0][1]}, a23 = {st[u][0][2], st[u][0][3]};
                    f2 b01 = {st[u][1][0], st[u][1][1]}, b23 = {st[u][1][2], st[u][1][3]};
                    f2 l0, l1, h0, h1;
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(l0) : "v"(a01), "v"(a23));
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(l1) : "v"(b01), "v"(b23));
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(h0) : "v"(a01), "v"(a23));
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(h1) : "v"(b01), "v"(b23));
                    f4 lo = {l0[0], l0[1], l1[0], l1[1]};
                    f4 hi = {h0[0], h0[1], h1[0], h1[1]};
                    *reinterpret_cast<f4*>(dst) = lo;
                    *reinterpret_cast<f4*>(dst + 4) = hi;
                } else if (FEAT & 8) {
                    f4 lo = {st[u][0][0], st[u][0][2], st[u][1][0], st[u][1][2]};
                    f4 hi = {st[u][0][1], st[u][0][3], st[u][1][1], st[u][1][3]};
                    *reinterpret_cast<f4*>(dst) = lo;
                    *reinterpret_cast<f4*>(dst + 4) = hi;
                } else {
                    *reinterpret_cast<f4*>(dst) = st[u][0];
                    *reinterpret_cast<f4*>(dst + 4) = st[u][1];
                }
            }
        }
        if (FEAT & 1) __syncthreads();
        if (FEAT & 4) {
            // the access pattern of a 3x3 conv over 128 channels at 135x240: per K-tile, 128 pixel rows x 128 B
            // (4 lanes share a 128-B line), 4 K-tiles per tap walk through one 512-B pixel record, taps re-touch
            // neighbouring pixels; weights: 128 rows of 1152 floats, L2 resident
            const int kt = it % 36, tap = kt >> 2, chunk = kt & 3;
            const unsigned npix = 8u * 135u * 240u;
            const unsigned tile = (blockIdx.x + (unsigned)(it / 36) * gridDim.x) % (npix / 128u);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int pix = (int)(tile * 128u) + (tid >> 2) + 64 * u + (tap / 3 - 1) * 240 + (tap % 3 - 1);
                pix = pix < 0 ? 0 : (pix >= (int)npix ? (int)npix - 1 : pix);
                const float* src = big + (size_t)pix * 128 + chunk * 32 + (tid & 3) * 8;
                st[u][0] = *reinterpret_cast<const f4*>(src);
                st[u][1] = *reinterpret_cast<const f4*>(src + 4);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float* src = big + (size_t)npix * 128 + (size_t)((tid >> 2) + 64 * u) * 1152 + kt * 32 + (tid & 3) * 8;
                st[2 + u][0] = *reinterpret_cast<const f4*>(src);
                st[2 + u][1] = *reinterpret_cast<const f4*>(src + 4);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            f4 af[2], bf[2];
            if (FEAT & 16) {
                // unpermuted rows: lane half h reads k = h, 2 + h, 4 + h, 6 + h of the octet (two ds_read2_b32)
                const int hh = lane >> 5;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* q = a_frag - hh * 4 + hh + i * 32 * STRIDE + o * 8;
                    af[i] = (f4){q[0], q[2], q[4], q[6]};
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float* q = b_frag - hh * 4 + hh + j * 32 * STRIDE + o * 8;
                    bf[j] = (f4){q[0], q[2], q[4], q[6]};
                }
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f4*>(a_frag + i * 32 * STRIDE + o * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f4*>(b_frag + j * 32 * STRIDE + o * 8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    for (int u = 0; u < 4; ++u) s += st[u][0][0] + st[u][1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Candidate K loop without register staging: v_mfma_f32_32x32x1_2b_f32 (K = 1 per instruction, the two lane halves
// are two 32-row blocks of the M tile sharing B) needs NO K permutation -- every lane consumes 4 consecutive k from one
// ds_read_b128 of the natural row-major layout -- so tiles can go global -> LDS directly (global_load_lds_dwordx4,
// lane-linear destination; bank conflicts avoided by an XOR swizzle of the 16-byte quad index with the row, applied
// to the SOURCE address and to the read), double-buffered, one barrier per K-tile.
template <int MODE>
__global__ __launch_bounds__(256, 2) void peak32_dlds(float* out, int iters, const float* big) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 256 * 36];  // 2 buffers x (128 A rows + 128 B rows) x 32 k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned npix = 8u * 135u * 240u;
    f16v acc[2][2];  // [n half][block pair as one 32-register accumulator split in two]
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    typedef float f32v __attribute__((ext_vector_type(32)));
    f32v c0, c1;
    for (int r = 0; r < 32; ++r) c0[r] = 0.f, c1[r] = 0.f;

    // per-slot constants: this thread's 8 (row, quad) slots of a tile; waves 0-1 stage A rows, waves 2-3 B rows
    int s_pix[8], s_q[8];
    const float* s_w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int slot = (wave * 8 + i) * 64 + lane;
        const int row = slot >> 3;
        s_q[i] = ((slot & 7) ^ (row & 7)) * 4;
        s_pix[i] = row;
        s_w[i] = big + (size_t)npix * 128 + (size_t)(row & 127) * 1152 + s_q[i];
    }
    auto issue = [&](int it, int buf) {
        const int kt = it % 36, tap = kt >> 2, chunk = kt & 3;
        const int tile = (int)((blockIdx.x + (unsigned)(it / 36) * gridDim.x) % (npix / 128u));
        const int toff = tile * 128 + (tap / 3 - 1) * 240 + (tap % 3 - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* src;
            if (wave < 2) {  // wave-uniform
                int pix = s_pix[i] + toff;
                pix = pix < 0 ? 0 : (pix >= (int)npix ? (int)npix - 1 : pix);
                src = big + (size_t)pix * 128 + chunk * 32 + s_q[i];
            } else {
                src = s_w[i] + kt * 32;
            }
            float* dst = smem + buf * (256 * 32) + (wave * 8 + i) * 256;  // wave-uniform base, + lane * 16 B implicit
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    issue(0, 0);
    __syncthreads();
    const int a_row = wm * 64 + lane;
    const int b_row0 = 128 + wn * 64 + (lane & 31);
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        if (MODE & 1) issue(it + 1, cur ^ 1);
        const float* base = smem + cur * (256 * 32);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int oa = (MODE & 4) ? o : (o ^ (a_row & 7)), ob = (MODE & 4) ? o : (o ^ (b_row0 & 7));
            const f4 a4 = *reinterpret_cast<const f4*>(base + a_row * ((MODE & 4) ? 36 : 32) + (oa << 2));
            const f4 b0 = *reinterpret_cast<const f4*>(base + b_row0 * ((MODE & 4) ? 36 : 32) + (ob << 2));
            const f4 b1 = *reinterpret_cast<const f4*>(base + (b_row0 + 32) * ((MODE & 4) ? 36 : 32) + (ob << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a4[e], b0[e], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x1f32(a4[e], b1[e], c1, 0, 0, 0);
            }
        }
        if (MODE & 2) __syncthreads();
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += c0[r] + c1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + acc[0][0][0];
}

template <int FEAT>
static double run_lds(float* out, const float* din, const float* big, unsigned mask, int cus, hipEvent_t e0, hipEvent_t e1) {
    float ms;
    const int it = 36 * 56, wpc = 2;
    hipLaunchKernelGGL(peak32_lds<FEAT>, dim3(cus * wpc), dim3(256), 0, 0, out, din, 10, big, mask);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(peak32_lds<FEAT>, dim3(cus * wpc), dim3(256), 0, 0, out, din, it, big, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    return 10.0 * 2.0 * 32 * 32 * 2 * 64.0 * it * 4.0 * cus * wpc / ms * 1e-9;
}

template <int K>
__global__ __launch_bounds__(256) void peak16(float* out, int iters, float a, float b) {
    f4v acc[K];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// One wave: D = A(16x4) * B(4x16) + C through one 16x16x4 instruction, `steps` times with fresh A/B.
__global__ void order16(const float* A, const float* B, const float* C, float* D, int steps) {
    const int l = threadIdx.x;
    f4v acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * (l / 16) + r) * 16 + (l % 16)];
    for (int s = 0; s < steps; ++s) {
        const float a = A[s * 64 + (l % 16) * 4 + (l / 16)];   // A[i][k], i = l%16, k = l/16
        const float b = B[s * 64 + (l / 16) * 16 + (l % 16)];  // B[k][j], k = l/16, j = l%16
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = acc[r];
}

static float rnd_wide() {
    const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
    const int e = rand() % 24 - 12;
    return ldexpf(m, e);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.gcnArchName, cus, prop.clockRate / 1000);
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int pass = 0; pass < 2; ++pass) {
        for (int wpc = 1; wpc <= 2; ++wpc) {  // workgroups of 4 waves per CU
            float ms;
            hipLaunchKernelGGL(peak32<4>, dim3(cus * wpc), dim3(256), 0, 0, out, 100, 1.f, 1.f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(peak32<4>, dim3(cus * wpc), dim3(256), 0, 0, out, iters, 1.f, 1e-9f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            double fl = 2.0 * 32 * 32 * 2 * 4.0 * iters * 4.0 * cus * wpc;
            if (pass) printf(", \"mfma32x32x2_f32_tflops_wg%d\": %.1f", wpc, fl / ms * 1e-9);
            hipEventRecord(e0);
            hipLaunchKernelGGL(peak16<4>, dim3(cus * wpc), dim3(256), 0, 0, out, iters, 1.f, 1e-9f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            fl = 2.0 * 16 * 16 * 4 * 4.0 * iters * 4.0 * cus * wpc;
            if (pass) printf(", \"mfma16x16x4_f32_tflops_wg%d\": %.1f", wpc, fl / ms * 1e-9);
        }
    }
    // long run (about 2 s) to see the sustained, power-managed clock
    {
        float ms;
        hipEventRecord(e0);
        for (int r = 0; r < 40; ++r)
            hipLaunchKernelGGL(peak32<4>, dim3(cus * 2), dim3(256), 0, 0, out, iters * 4, 1.f, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 40.0 * 2.0 * 32 * 32 * 2 * 4.0 * iters * 4 * 4.0 * cus * 2;
        printf(", \"mfma32x32x2_f32_tflops_sustained\": %.1f, \"sustained_s\": %.2f", fl / ms * 1e-9, ms * 1e-3);
    }

    {
        std::vector<float> h(4096);
        srand(3);
        for (auto& v : h) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 1e-3f;
        float* din;
        hipMalloc(&din, 4096 * 4);
        hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        float ms;
        hipLaunchKernelGGL(peak32_rand, dim3(cus * 2), dim3(256), 0, 0, out, din, 100);
        hipEventRecord(e0);
        for (int r = 0; r < 40; ++r)
            hipLaunchKernelGGL(peak32_rand, dim3(cus * 2), dim3(256), 0, 0, out, din, iters * 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 40.0 * 2.0 * 32 * 32 * 2 * 4.0 * iters * 4 * 4.0 * cus * 2;
        printf(", \"mfma32x32x2_f32_tflops_random_operands\": %.1f, \"random_s\": %.2f", fl / ms * 1e-9, ms * 1e-3);
    }
    {
        std::vector<float> h(4096);
        srand(5);
        for (auto& v : h) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        float* din;
        hipMalloc(&din, 4096 * 4);
        hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        float* big;
        const unsigned big_elems = 1u << 26;  // 256 MB
        hipMalloc(&big, (size_t)big_elems * 4);
        hipMemset(big, 0, (size_t)big_elems * 4);
        const unsigned mask = (big_elems - 1) & ~31u;
        printf(", \"lds_fed_loop_tflops\": {\"mfma_only\": %.1f", run_lds<0>(out, din, big, mask, cus, e0, e1));
        printf(", \"+barriers\": %.1f", run_lds<1>(out, din, big, mask, cus, e0, e1));
        printf(", \"+barriers+ds_write\": %.1f", run_lds<3>(out, din, big, mask, cus, e0, e1));
        printf(", \"+barriers+ds_write+perm\": %.1f", run_lds<11>(out, din, big, mask, cus, e0, e1));
        printf(", \"+global_loads\": %.1f", run_lds<4>(out, din, big, mask, cus, e0, e1));
        printf(", \"+barriers+ds_write+global_loads\": %.1f", run_lds<7>(out, din, big, mask, cus, e0, e1));
        printf(", \"all\": %.1f", run_lds<15>(out, din, big, mask, cus, e0, e1));
        printf(", \"all_pk_mov\": %.1f", run_lds<7 + 32>(out, din, big, mask, cus, e0, e1));
        {
            auto run = [&](auto kern) {
                float ms;
                const int it = 36 * 56;
                hipLaunchKernelGGL(kern, dim3(cus * 2), dim3(256), 0, 0, out, 10, big);
                hipEventRecord(e0);
                for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(cus * 2), dim3(256), 0, 0, out, it, big);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                return 10.0 * 4096.0 * 64.0 * it * 4.0 * cus * 2 / ms * 1e-9;
            };
            printf(", \"32x32x1_2b\": {\"lds_only_swizzled\": %.1f", run(peak32_dlds<0>));
            printf(", \"lds_only_padded_rows\": %.1f", run(peak32_dlds<4>));
            printf(", \"+barrier\": %.1f", run(peak32_dlds<2>));
            printf(", \"+direct_to_lds_loads\": %.1f", run(peak32_dlds<1>));
            printf(", \"+both\": %.1f}", run(peak32_dlds<3>));
        }
        printf(", \"mfma_only_read2\": %.1f", run_lds<16>(out, din, big, mask, cus, e0, e1));
        printf(", \"all_noperm_read2\": %.1f}", run_lds<16 + 7>(out, din, big, mask, cus, e0, e1));
    }
    // order check
    const int steps = 16, trials = 200;
    std::vector<float> A(steps * 64), B(steps * 64), C(256), D(256);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dC, 1024);
    hipMalloc(&dD, 1024);
    long bad_chain = 0, bad_pair = 0, total = 0;
    srand(7);
    for (int t = 0; t < trials; ++t) {
        for (auto& v : A) v = rnd_wide();
        for (auto& v : B) v = rnd_wide();
        for (auto& v : C) v = (t & 1) ? 0.f : rnd_wide();
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(order16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, steps);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float chain = C[i * 16 + j];
                for (int s = 0; s < steps; ++s)
                    for (int k = 0; k < 4; ++k) chain = fmaf(A[s * 64 + i * 4 + k], B[s * 64 + k * 16 + j], chain);
                // alternative: pairwise (k0*k1 summed first) to show the check discriminates
                float alt = C[i * 16 + j];
                for (int s = 0; s < steps; ++s) {
                    float p = 0.f;
                    for (int k = 0; k < 4; ++k) p = fmaf(A[s * 64 + i * 4 + k], B[s * 64 + k * 16 + j], p);
                    alt += p;
                }
                ++total;
                if (memcmp(&chain, &D[i * 16 + j], 4)) ++bad_chain;
                if (memcmp(&alt, &D[i * 16 + j], 4)) ++bad_pair;
            }
    }
    printf(", \"order16x16x4\": {\"checked\": %ld, \"mismatch_vs_ordered_fmaf_chain\": %ld, \"mismatch_vs_blockwise_sum\": %ld}}\n",
           total, bad_chain, bad_pair);
    return 0;
}
