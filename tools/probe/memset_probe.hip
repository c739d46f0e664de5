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
// (the library's launches carry ~450 bytes of arguments by value -- BwdArgs + SplitK: `Big` makes the probe's as large, with
//  recognisable words, in case the replayed fill takes its pattern from memory a neighbouring node's arguments occupy)
struct Big { int v[112]; };
__global__ void write_some(float* scratch, long long stride, int Z, long long span, int salt, Big big) {
    if (big.v[5] == 12345678) scratch[0] = 1.0f;                 // (keeps the argument alive)
    spin(5000);                                                  // 100 MHz clock: 50 us
    // slice z writes every position except those with (i % 97) == 3 (the "padding" no kernel writes)
    const int z = blockIdx.y;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < span; i += (long long)gridDim.x * blockDim.x)
        if (i % 97 != 3) scratch[z * stride + i] = 1.0f + (float)((i + salt) & 7);
}
__global__ void reduce(const float* scratch, long long stride, int Z, long long span, float* out, Big big) {
    if (big.v[7] == 12345678) out[0] = 1.0f;
    spin(5000);
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

// `between`: what the host does between two launches of the graph (the trainer polls a few words of device memory, torch fills and
// copies small tensors): 0 nothing | 1 an eager 4-byte-pattern memset of another buffer | 2 a 12-byte copy to pageable host memory |
// 3 the same to pinned memory, asynchronously, on another stream | 4 all of them
__global__ void small_kernel(float* p, int a, int b, int c, long long d) { if (threadIdx.x == 0 && blockIdx.x == 0) p[a] = (float)(b + c) + (float)d; }
// `passes` backwards per graph with `extra` small kernel nodes after each (the trainer's 16-iteration window holds ~50 memset
// nodes among ~500 kernel nodes)
// `copies`: small device-to-device hipMemcpyAsync calls (memcpy NODES: torch's copy_ under capture) after each pass
int run(bool graph, bool memset_node, int replays, long long offset_floats = 0, int between = 0, int passes = 2, int extra = 0, int copies = 0) {
    const int Z = 256;
    const long long span = 34564, stride = 34564;
    float *scratch_base, *scratch, *out;
    // (PyTorch's caching allocator hands out 512-byte-aligned pieces of larger segments: `offset_floats` moves the destination off
    //  the allocation's own alignment)
    CK(hipMalloc(&scratch_base, sizeof(float) * (Z * stride + 4096)));
    scratch = scratch_base + offset_floats;
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
    Big big;
    for (int i = 0; i < 112; ++i) big.v[i] = 0x1000 + i;
    auto body = [&]() {
        for (int pass = 0; pass < passes; ++pass) {
            if (memset_node) CK(hipMemsetAsync(other, 0x7f - pass, other_bytes, s));
            if (memset_node) CK(hipMemsetAsync(scratch, 0, sizeof(float) * Z * stride, s));
            else hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, s, (float4*)scratch, (long long)Z * stride / 4);   // (16-byte aligned offsets only)
            hipLaunchKernelGGL(write_some, dim3(64, Z), dim3(256), 0, s, scratch, stride, Z, span, pass, big);
            hipLaunchKernelGGL(reduce, dim3(136), dim3(256), 0, s, scratch, stride, Z, span, out, big);
            for (int c = 0; c < copies; ++c)
                CK(hipMemcpyAsync(other + 4096 + 64 * c, other + 8192 + 64 * c, 12 + 4 * (c & 3), hipMemcpyDeviceToDevice, s));
            for (int e = 0; e < extra; ++e) hipLaunchKernelGGL(small_kernel, dim3(4), dim3(64), 0, s, (float*)other, 0, 1, 12, (long long)e);
        }
    };
    if (graph) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));   // (torch captures in relaxed / thread-local mode)
        body();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        unsigned int *small_dev, *pinned, pageable[4];
        hipStream_t s2;
        CK(hipStreamCreate(&s2));
        CK(hipMalloc(&small_dev, 4096));
        CK(hipHostMalloc(&pinned, 4096));
        for (int r = 0; r < replays; ++r) {
            CK(hipGraphLaunch(ge, s));
            if (between == 1 || between == 4) CK(hipMemsetD32Async((hipDeviceptr_t)small_dev, 0x0C0C0C0Cu + r, 1024, s));
            if (between == 2 || between == 4) CK(hipMemcpy(pageable, small_dev, 12, hipMemcpyDeviceToHost));
            if (between == 3 || between == 4) { CK(hipMemcpyAsync(pinned, small_dev, 12, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); }
        }
        CK(hipStreamSynchronize(s));
        CK(hipFree(small_dev)); CK(hipHostFree(pinned)); CK(hipStreamDestroy(s2));
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
            double want = 0;
            for (int pass = 0; pass < passes; ++pass) want += (double)replays * Z * (1.0 + ((i + pass) & 7));
            if (h[i] != (float)want) expect_err += 1;
        }
    }
    if (memset_node) {
        std::vector<unsigned char> ho(other_bytes);
        CK(hipMemcpy(ho.data(), other, other_bytes, hipMemcpyDeviceToHost));
        long long wrong = 0;
        for (size_t i = 0; i < other_bytes; ++i) wrong += ho[i] != (unsigned char)(0x7f - (passes - 1)) && i >= 16384;
        if (wrong) { printf("   second memset's buffer: %lld of %zu bytes are not its last pattern\n", wrong, other_bytes); ++bad; }
    }
    CK(hipFree(other));
    if (between) printf("(between the launches: %d) ", between);
    printf("%-6s %-12s %3d replays: %d never-written positions non-zero, %g written positions off\n", graph ? "graph" : "eager",
           memset_node ? "memset" : "zero kernel", replays, bad, expect_err);
    CK(hipFree(scratch_base)); CK(hipFree(out));
    return bad;
}

extern "C" int memset_probe_main() {
    int bad = 0;
    for (int rep = 0; rep < 2; ++rep) {
        bad += run(false, true, 20);
        bad += run(true, false, 20);
        bad += run(true, true, 20);
        for (long long off : {128ll, 132ll, 1024ll, 3ll * 128}) {   // 512-byte / 528-byte / 4-KB-off / 1.5-KB destinations
            printf("destination offset %lld floats: ", off);
            bad += run(true, true, 20, off);
        }
        for (int between = 1; between <= 4; ++between) bad += run(true, true, 20, 0, between);
        printf("48 passes per graph, 8 small kernels after each: ");
        bad += run(true, true, 6, 0, 4, 48, 8);
        printf("2 passes, 3 small device-to-device copies after each: ");
        bad += run(true, true, 20, 0, 0, 2, 0, 3);
        printf("48 passes, 8 small kernels and 3 small device-to-device copies after each: ");
        bad += run(true, true, 6, 0, 4, 48, 8, 3);
    }
    return bad ? 1 : 0;
}


// The same launches on a stream and buffers the CALLER owns (tools/probe/memset_probe.py: torch tensors, captured by
// torch.cuda.graph into torch's private pool, replayed by torch).
extern "C" int memset_probe_body(void* stream, float* scratch, float* out, unsigned char* other, long long other_bytes, int Z,
                                 long long stride, long long span, int passes, int memset_node) {
    hipStream_t s = (hipStream_t)stream;
    Big big;
    for (int i = 0; i < 112; ++i) big.v[i] = 0x1000 + i;
    for (int pass = 0; pass < passes; ++pass) {
        if (memset_node && other) CK(hipMemsetAsync(other, 0x7f - pass, other_bytes, s));
        if (memset_node) CK(hipMemsetAsync(scratch, 0, sizeof(float) * Z * stride, s));
        else hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, s, (float4*)scratch, (long long)Z * stride / 4);
        hipLaunchKernelGGL(write_some, dim3(64, Z), dim3(256), 0, s, scratch, stride, Z, span, pass, big);
        hipLaunchKernelGGL(reduce, dim3(136), dim3(256), 0, s, scratch, stride, Z, span, out, big);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- The same chain with the library's ARGUMENT LAYOUTS (round 6, after tools/probe/memset_backward_probe.py reproduced the
// stale fill with one captured rpo_mlp_backward call and the pattern turned out to be bytes 32..47 of the reduce kernel's
// arguments -- {stride, gradmax pointer}): td(88 bytes + int) -> memset -> rows(384 + 40 bytes) -> weights(384 + 40) ->
// reduce(40 bytes + pointer), 64 floats per slice that no kernel writes.
struct S40 { float* scratch; float* lo; long long span; int Z; long long stride; };
struct A384 { int v[96]; };
struct T88 { int v[22]; };
__global__ void k_td(T88 t, int n) { if (t.v[3] == 12345678 && n == -1) __builtin_trap(); }
__global__ void k_rows(A384 a, S40 k) {
    const int z = blockIdx.y;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < k.span; i += (long long)gridDim.x * blockDim.x)
        k.scratch[z * k.stride + i] = 1.0f + (float)((i + a.v[0]) & 7);
}
__global__ void k_weights(A384 a, S40 k) { if (a.v[5] == 12345678 && k.Z == -1) __builtin_trap(); }
__global__ void k_reduce(S40 k, float* out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < k.span; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int z = 0; z < k.Z; ++z) s += k.scratch[z * k.stride + i];
        if (s != 0.0f) out[i] += s;
    }
}
extern "C" int memset_probe_layouts(void* stream_or_null, int replays) {
    const int Z = 256;
    const long long span = 34508, stride = span + 64;
    float *scratch, *out;
    CK(hipMalloc(&scratch, sizeof(float) * Z * stride));
    CK(hipMalloc(&out, sizeof(float) * span));
    CK(hipMemset(scratch, 0, sizeof(float) * Z * stride));
    CK(hipMemset(out, 0, sizeof(float) * span));
    hipStream_t s = (hipStream_t)stream_or_null;
    if (!s) CK(hipStreamCreate(&s));
    A384 a; T88 t;
    for (int i = 0; i < 96; ++i) a.v[i] = 0x2000 + i;
    for (int i = 0; i < 22; ++i) t.v[i] = 0x3000 + i;
    S40 k{scratch, out, span, Z, stride};
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_td, dim3(4096), dim3(256), 0, s, t, 1 << 20);
    CK(hipMemsetAsync(scratch, 0, sizeof(float) * Z * stride, s));
    hipLaunchKernelGGL(k_rows, dim3(64, Z), dim3(256), 0, s, a, k);
    hipLaunchKernelGGL(k_weights, dim3(256), dim3(512), 0, s, a, k);
    hipLaunchKernelGGL(k_reduce, dim3(541), dim3(256), 0, s, k, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<int> h((size_t)Z * stride);
    CK(hipMemcpy(h.data(), scratch, sizeof(float) * Z * stride, hipMemcpyDeviceToHost));
    long long bad = 0; int shown = 0;
    for (int z = 0; z < Z; ++z)
        for (long long i = span; i < stride; ++i)
            if (h[z * stride + i] != 0) { if (shown++ < 1) printf("   slice %d tail: {%d, %d, %d, %d}\n", z, h[z * stride + span], h[z * stride + span + 1], h[z * stride + span + 2], h[z * stride + span + 3]); ++bad; }
    printf("library argument layouts, %d replays: %lld of %lld never-written words are not zero\n", replays, bad, (long long)Z * 64);
    CK(hipFree(scratch)); CK(hipFree(out));
    return bad ? 1 : 0;
}

int main() { int rc = memset_probe_main(); rc |= memset_probe_layouts(nullptr, 6); return rc; }
