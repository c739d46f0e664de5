// Dispatch-cost probe: a kernel that does (almost) nothing, with the resource footprint of the column-split slab
// workgroups (13 KB LDS, ~100 VGPRs), launched with the same number of THREADS in different workgroup shapes.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/dispatch_probe.hip -o tools/probe/libdispatch_probe.so
#include <hip/hip_runtime.h>

template <int T>
__global__ __launch_bounds__(T) void probe_kernel(float* out, int work) {
    __shared__ float lds[13 * 256];
    float acc[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i] = (float)(threadIdx.x + i);
    for (int w = 0; w < work; ++w) {
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += acc[i];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0 && lds[1] == -1.0f) out[blockIdx.x] = lds[0];
}

extern "C" int probe_launch(int threads_per_wg, int wgs, int work, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (threads_per_wg) {
        case 64: hipLaunchKernelGGL(probe_kernel<64>, dim3(wgs), dim3(64), 0, s, out, work); break;
        case 128: hipLaunchKernelGGL(probe_kernel<128>, dim3(wgs), dim3(128), 0, s, out, work); break;
        case 256: hipLaunchKernelGGL(probe_kernel<256>, dim3(wgs), dim3(256), 0, s, out, work); break;
        case 512: hipLaunchKernelGGL(probe_kernel<512>, dim3(wgs), dim3(512), 0, s, out, work); break;
        case 1024: hipLaunchKernelGGL(probe_kernel<1024>, dim3(wgs), dim3(1024), 0, s, out, work); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
