// Timestamped spin kernels for tools/probe/branch_probe.py: what does a fork / join between two captured branches of a hipGraph
// cost on this stack?  Each launch records the device clock (100 MHz) at its start and end.
#include <hip/hip_runtime.h>

__global__ void spin_kernel(long long* stamps, int slot, int spin_ticks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long t0 = wall_clock64();
        stamps[2 * slot] = t0;
        while (wall_clock64() - t0 < spin_ticks) {}
        stamps[2 * slot + 1] = wall_clock64();
    }
}

extern "C" int probe_spin(long long* stamps, int slot, int spin_ticks, int wgs, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream, stamps, slot, spin_ticks);
    return (int)hipGetLastError();
}
