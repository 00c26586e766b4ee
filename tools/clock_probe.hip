// Tuning aid: one wavefront spins for `spin_us` and reports the shader clock it saw
// (s_memtime cycles per s_memrealtime 100 MHz tick).  Launched on a side stream while the kernels under
// study run, it tells at which frequency the chip actually executed them.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/clock_probe.hip -o /tmp/libclock_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void spin_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r = r0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r = wall_clock64();
    }
    out[0] = clock64() - c0;
    out[1] = r - r0;
}

extern "C" __attribute__((visibility("default"))) int clock_probe_launch(void* out2, double spin_us, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out2,
                       (unsigned long long)(spin_us * 100.0));
    return (int)hipGetLastError();
}
