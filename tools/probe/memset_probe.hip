// Stand-alone reproducer attempt for DESIGN.md 4.5 (round 6): does a hipMemsetAsync captured into a hipGraph (a memset NODE) zero
// what the eager call zeroes?  Pattern of the large-batch backward: [memset scratch] -> [kernel A: every slice writes SOME
// positions of its copy] -> [kernel R: sums the copies into out where the sum is non-zero], twice per graph (two backwards share
// the scratch), replayed back to back.  A position that no kernel writes must stay zero in `out`.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/memset_probe tools/probe/memset_probe.hip && tools/probe/memset_probe      (ROCm's runtime)
//   python tools/probe/memset_probe.py   (the same code inside a Python process bound to the HIP runtime PyTorch bundles)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ void spin(long long cycles) {                        // (long-running kernels, like the 500 us backward launches)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
__global__ void write_some(float* scratch, long long stride, int Z, long long span, int salt) {
    spin(20000);                                                 // 100 MHz clock: 200 us
    // slice z writes every position except those with (i % 97) == 3 (the "padding" no kernel writes)
    const int z = blockIdx.y;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < span; i += (long long)gridDim.x * blockDim.x)
        if (i % 97 != 3) scratch[z * stride + i] = 1.0f + (float)((i + salt) & 7);
}
__global__ void reduce(const float* scratch, long long stride, int Z, long long span, float* out) {
    spin(20000);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < span; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int z = 0; z < Z; ++z) s += scratch[z * stride + i];
        if (s != 0.0f) out[i] += s;
    }
}
__global__ void zero_kernel(float4* p, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int run(bool graph, bool memset_node, int replays) {
    const int Z = 256;
    const long long span = 34564, stride = 34564;
    float *scratch, *out;
    CK(hipMalloc(&scratch, sizeof(float) * Z * stride));
    CK(hipMalloc(&out, sizeof(float) * span));
    CK(hipMemset(scratch, 0, sizeof(float) * Z * stride));
    CK(hipMemset(out, 0, sizeof(float) * span));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // a SECOND kind of memset in the same graph (another destination, another size, another value): if the nodes' arguments
    // were mixed up at replay, `scratch` would show this pattern, or `other` zeros
    unsigned char* other;
    const size_t other_bytes = 3 * 1000 * 1000 + 13;
    CK(hipMalloc(&other, other_bytes));
    CK(hipMemset(other, 0, other_bytes));
    auto body = [&]() {
        for (int pass = 0; pass < 2; ++pass) {
            if (memset_node) CK(hipMemsetAsync(other, 0x7f - pass, other_bytes, s));
            if (memset_node) CK(hipMemsetAsync(scratch, 0, sizeof(float) * Z * stride, s));
            else hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, s, (float4*)scratch, (long long)Z * stride / 4);
            hipLaunchKernelGGL(write_some, dim3(64, Z), dim3(256), 0, s, scratch, stride, Z, span, pass);
            hipLaunchKernelGGL(reduce, dim3(136), dim3(256), 0, s, scratch, stride, Z, span, out);
        }
    };
    if (graph) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));   // (torch captures in relaxed / thread-local mode)
        body();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
    } else {
        for (int r = 0; r < replays; ++r) body();
    }
    CK(hipStreamSynchronize(s));
    std::vector<float> h(span);
    CK(hipMemcpy(h.data(), out, sizeof(float) * span, hipMemcpyDeviceToHost));
    int bad = 0; double expect_err = 0;
    for (long long i = 0; i < span; ++i) {
        if (i % 97 == 3) { if (h[i] != 0.0f) { if (bad < 3) printf("   position %lld (never written) = %g\n", i, h[i]); ++bad; } }
        else {
            const double want = (double)replays * Z * ((1.0 + ((i + 0) & 7)) + (1.0 + ((i + 1) & 7)));
            if (h[i] != (float)want) expect_err += 1;
        }
    }
    if (memset_node) {
        std::vector<unsigned char> ho(other_bytes);
        CK(hipMemcpy(ho.data(), other, other_bytes, hipMemcpyDeviceToHost));
        long long wrong = 0;
        for (size_t i = 0; i < other_bytes; ++i) wrong += ho[i] != 0x7e;
        if (wrong) { printf("   second memset's buffer: %lld of %zu bytes are not its last pattern\n", wrong, other_bytes); ++bad; }
    }
    CK(hipFree(other));
    printf("%-6s %-12s %3d replays: %d never-written positions non-zero, %g written positions off\n", graph ? "graph" : "eager",
           memset_node ? "memset" : "zero kernel", replays, bad, expect_err);
    CK(hipFree(scratch)); CK(hipFree(out));
    return bad;
}

extern "C" int memset_probe_main() {
    int bad = 0;
    for (int rep = 0; rep < 2; ++rep) {
        bad += run(false, true, 20);
        bad += run(true, false, 20);
        bad += run(true, true, 20);
    }
    return bad ? 1 : 0;
}
int main() { return memset_probe_main(); }
