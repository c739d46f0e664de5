// Large-batch forward of the 128 -> 256 scalar-head networks: WEIGHTS STATIONARY IN LDS, ROWS STREAM THROUGH, WAVES INDEPENDENT,
// EVERYTHING TRANSPOSED (round 5; reference work: rpo/algo/rpo_ddpg.py:163-205 at SURVEY 8d-iii's batch of 256 * N rows).
//
// The 64-row tile kernel (mlp_tile.h, RT = 4) gives every tile a workgroup that pulls W0 (128 KB) from L2 again, splits the
// hidden COLUMNS over its eight waves and meets at four barriers per tile: 0.63 of the f32 MFMA peak at 2^20 rows.  Here one
// persistent workgroup per CU stages W0 ONCE into LDS (256 x 136 floats = 136 KB of the CU's 160 KB; row stride 136: the
// ds_read_b128 of lane (li, lg) at W0[16 jt + li][16 it + 4 lg ..+3] is conflict-free for 136 / 4 = 2 mod 16; the padding 132 of
// the row-tile kernels has a two-way conflict) and every WAVE owns whole 16-row tiles end to end: no barrier after the staging,
// no LDS traffic between waves.
//
// What the first version of this kernel taught (timing-only builds with phases left out, RPO_STREAM_SKIP): v_mfma_f32_16x16x4_f32
// runs at the f32 VECTOR rate because it runs on the vector FMA lanes -- vector instructions of the other waves of a SIMD do
// not hide under it, their issue time ADDS to the MFMA time (MFMA loop alone 0.94 of the peak; + 1300 vector instructions per
// tile for the first layer, head and stores: 0.73).  So this version has almost no vector instructions, by computing both
// layers TRANSPOSED:
//   * first layer  x0^T[e][row] = Wfl[e][u] in^T[u][row] + bias[e]: three MFMAs per 16 e (K = 12: the <= 11 inputs, zero
//     padded), the bias as the accumulator's initial value.  Its C layout -- lane (li, lg) holds x0[row li][16 it + 4 lg + i] --
//     IS the B-operand layout the hidden layer needs for k-group it (B[k = lg][n = li]), so relu(x0) feeds the next MFMAs from
//     the registers it lands in, and the pre-activation store is one 16-byte store per lane;
//   * hidden layer h1^T[j][row] = W0[j][e] x1^T[e][row]: A from LDS (the same ds_read_b128 as before), 16 accumulators of
//     the row tile; lane (li, lg) ends with h1[row li][16 jt + 4 lg + i]: 16-byte pre-activation stores, and the head is a
//     lane-local fma chain + two cross-lane adds per partial.
// Arithmetic = the row-tile kernels' to the bit (tests/test_trainer_gpu.py, tools/probe_mlp_large.py check): an MFMA is the
// k-ordered fmaf chain the vector first layer was (bias, state inputs, action inputs; products commute); the hidden layer's
// chains are unchanged (operands swapped); the head adds the old eight 32-column partials in the old association -- in the
// old kernel lane l' of a wave held fma(hr[32 w + 16 + l'], w1, hr[32 w + l'] w1) and a quad / row butterfly added the 16
// lanes as ((q0 + q1) + (q2 + q3)) + ... ; here lane (li, lg) holds the four q of lanes 4 lg .. 4 lg + 3 of its row.
// (Inputs beyond S + A are zero-weighted zeros: fmaf(0, 0, x) == x except for x == -0.0 -> +0.0, a sign of zero.)
#pragma once
#include <type_traits>

#include "mlp_tile.h"

namespace rpo_mlp_dev {

#ifndef RPO_STREAM_SKIP
#define RPO_STREAM_SKIP 0          // debug builds only (tools/probe/build_stream_variants.sh): 1 no input loads, 2 no first layer,
#endif                             // 4 no epilogue -- wrong results, timing only
#ifndef RPO_STREAM_NT
#define RPO_STREAM_NT 1            // pre-activation stores: 1 non-temporal (default), 0 plain (A/B builds)
#endif
#if RPO_STREAM_NT
#define RPO_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define RPO_STREAM_STORE(v, p) (*(p) = (v))
#endif
constexpr int kStreamLdW = 136;        // floats between rows of W0 in LDS
constexpr int kStreamIn = 11;          // S + A <= 11 inputs (K = 12 of the first-layer MFMAs, the last slot is zero)

template <int H>
struct StreamLds {
    __attribute__((aligned(16))) float w0[H * kStreamLdW];
    __attribute__((aligned(16))) float fl[128 * 16];            // [e][lg][ks]: W_first[e][4 ks + lg] (ks = 0..2; slot 3 unused)
    __attribute__((aligned(16))) float bias[128];               // bs (+ ba): the accumulator's initial value
    __attribute__((aligned(16))) float b0[H], w1a[H], w1b[H];
};

static inline bool stream_shape_ok(const Mlp& net) {
    return !net.cat && net.E == 128 && net.H == 256 && net.hd <= 1 && net.S + net.A <= kStreamIn && net.S > 0;
}

__device__ __forceinline__ float stream_xlane(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// One 16-row tile, start to finish, by one wave.  FULL: every row of the tile exists (no guards: a guarded store is a branch
// of its own); otherwise the inputs come from a clamped row and the stores are guarded.  CH: column tiles per chunk of the
// hidden layer (an accumulator is revisited every CH MFMAs = 64+ cycles >= the 40 of a dependent pair).  SAVE / TWO: whether
// the pre-activations are stored / the network has a second head, as compile-time facts (1 / 0) on the hot path -- as run-time
// conditions (-1: the rare partial tile) every share of the epilogue is a branch region of its own.
// The B operand of the first layer for the tile at row0: in^T[u = 4 ks + lg][row li], three values per lane (rows beyond n:
// a clamped row -- their results are never stored).  Requested one tile AHEAD by the caller: at the tile's start the loads
// would expose a full memory latency per tile.
template <class ARGS>
__device__ __forceinline__ void stream_inputs(const ARGS& p, long long row0, int lane, float (&in3)[3]) {
    const Mlp& net = p.net;
    const int li = lane & 15, lg = lane >> 4, nin = net.S + net.A;
    const long long row = row0 + li, rc = row < p.n ? row : (long long)p.n - 1;
    const float* sp = p.s + (size_t)rc * p.s_stride;
    const float* ap = net.A > 0 ? p.a + (size_t)rc * p.a_stride : sp;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int u = 4 * ks + lg;
        const bool is_s = u < net.S, is_a = !is_s && u < nin;
        const float* q = is_s ? sp + u : (is_a ? ap + (u - net.S) : sp);
        const float v = (RPO_STREAM_SKIP & 1) ? 0.25f * (float)(u + li) : *q;
        in3[ks] = (is_s || is_a) ? v : 0.0f;
    }
}

// EMIT: what happens to a row's outputs -- emit(row, o0, o1, two) is called by the lane that holds row li's values (lg == 0) for
// every existing row: the forward kernels store them, the streaming rollout (fused.hip) keeps them for its per-lane phase.
template <int H, bool FULL, int CH, int SAVE, int TWO, class ARGS, class EMIT>
__device__ __forceinline__ void stream_tile(const ARGS& p, const StreamLds<H>& lds, int row0, int lane, float b1a, float b1b,
                                            const float (&in3)[3], const EMIT& emit) {
    constexpr int EIN = 128;
    const Mlp& net = p.net;
    const int li = lane & 15, lg = lane >> 4;
    const bool two = TWO < 0 ? net.n_out > 1 : TWO != 0;
    const bool save_x0 = SAVE < 0 ? p.x0_save != nullptr : SAVE != 0, save_h1 = SAVE < 0 ? p.h1_save != nullptr : SAVE != 0;
    const int row = row0 + li;
    const bool live = FULL || row < p.n;
    f32x4 acc[H / 16];
#pragma unroll
    for (int c = 0; c < H / 16; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float* x0p = (save_x0 && live) ? p.x0_save + (size_t)row * EIN + lg * 4 : nullptr;
#pragma unroll 1
    for (int it = 0; it < EIN / 16; ++it) {                      // (rolled: fully unrolled, the scheduler hoisted every LDS read of
        // ---- first layer, transposed: x0[row li][16 it + 4 lg + i] in the C layout               the tile and spilled 700 registers)
        f32x4 x0 = *reinterpret_cast<const f32x4*>(&lds.bias[it * 16 + lg * 4]);
        if (!(RPO_STREAM_SKIP & 2)) {
            const float4 w3 = *reinterpret_cast<const float4*>(&lds.fl[((it * 16 + li) * 4 + lg) * 4]);
            x0 = mfma4(w3.x, in3[0], x0);
            x0 = mfma4(w3.y, in3[1], x0);
            x0 = mfma4(w3.z, in3[2], x0);
        }
        if ((FULL && SAVE > 0) || x0p) RPO_STREAM_STORE(x0, reinterpret_cast<f32x4*>(x0p + it * 16));
        float a4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a4[m] = rpo_relu_bits(x0[m]);
        // ---- hidden layer, transposed: CH column tiles at a time, consecutive MFMAs on different accumulators; the A operands
        // (W0 from LDS) of chunk cc + 1 are requested BEFORE the MFMAs of chunk cc (register ping-pong; the scheduling barriers pin
        // that order: left alone the compiler issues the reads behind the chunk's last MFMAs, or -- unrolled -- all of them up front)
        const float* wb = &lds.w0[li * kStreamLdW + it * 16 + lg * 4];
        float4 bcur[CH], bnxt[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) bcur[q] = *reinterpret_cast<const float4*>(wb + 16 * q * kStreamLdW);
#pragma unroll
        for (int cc = 0; cc < H / 16; cc += CH) {
            if (cc + CH < H / 16) {
#pragma unroll
                for (int q = 0; q < CH; ++q) bnxt[q] = *reinterpret_cast<const float4*>(wb + 16 * (cc + CH + q) * kStreamLdW);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(bcur[q].x, a4[0], acc[cc + q]);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(bcur[q].y, a4[1], acc[cc + q]);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(bcur[q].z, a4[2], acc[cc + q]);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(bcur[q].w, a4[3], acc[cc + q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CH; ++q) bcur[q] = bnxt[q];
        }
    }
    if (RPO_STREAM_SKIP & 4) {
        float t = 0.0f;
#pragma unroll
        for (int c = 0; c < H / 16; ++c) t += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
        if (t == 12345.678f) p.out[row0] = t;
        return;
    }
    // ---- epilogue: acc[jt][i] = h1[row li][16 jt + 4 lg + i] before the bias
    float v0 = b1a, v1 = b1b;
    // ONE base register for the 32 - 48 table reads of the epilogue (b0 | w1a | w1b are contiguous: immediate offsets).  Opaque to
    // the optimiser on purpose: it otherwise forms one `base | constant` address per read outside the tile loop -- 48 registers
    // that it then spills around the MFMA loop and reloads here.
    typedef const f32x4 __attribute__((address_space(3))) * lds_f4;
    typedef const float __attribute__((address_space(3))) * lds_f;
    unsigned ep = (unsigned)(__UINTPTR_TYPE__)((lds_f)(&lds.b0[0])) + lg * 16;
    asm volatile("" : "+v"(ep));
    float* h1p = (save_h1 && live) ? p.h1_save + (size_t)row * H + lg * 4 : nullptr;
#pragma unroll
    for (int w8 = 0; w8 < H / 32; ++w8) {                        // the 32-column shares of the row-tile kernels' eight waves
        float q0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int jt = 2 * w8; jt < 2 * w8 + 2; ++jt) {
            const f32x4 b0 = *(lds_f4)(__UINTPTR_TYPE__)(ep + 64 * jt);
            const f32x4 wa = *(lds_f4)(__UINTPTR_TYPE__)(ep + 64 * jt + 4 * H);
            const f32x4 h = acc[jt] + b0;
            if ((FULL && SAVE > 0) || h1p) RPO_STREAM_STORE(h, reinterpret_cast<f32x4*>(h1p + 16 * jt));
#pragma unroll
            for (int i = 0; i < 4; ++i) q0[i] = fmaf(rpo_relu_bits(h[i]), wa[i], q0[i]);
            if (two) {
                const f32x4 wb2 = *(lds_f4)(__UINTPTR_TYPE__)(ep + 64 * jt + 8 * H);
#pragma unroll
                for (int i = 0; i < 4; ++i) q1[i] = fmaf(rpo_relu_bits(h[i]), wb2[i], q1[i]);
            }
        }
        // the old butterfly over the 16 lanes of a row, regrouped: quad sums are lane-local, the two row_shl steps are the two
        // exchanges below (lane lg = 0 adds in exactly the old order; the others get the same bits by commutativity)
        float s = (q0[0] + q0[1]) + (q0[2] + q0[3]);
        s = s + stream_xlane(s, lane ^ 16);
        s = s + stream_xlane(s, lane ^ 32);
        v0 += s;
        if (two) {
            float t = (q1[0] + q1[1]) + (q1[2] + q1[3]);
            t = t + stream_xlane(t, lane ^ 16);
            t = t + stream_xlane(t, lane ^ 32);
            v1 += t;
        }
        __builtin_amdgcn_sched_barrier(0);                       // (else every share's LDS reads are hoisted to the front: spills)
    }
    if (lg == 0 && live) {
        float o0 = v0;
        if (p.out_mode == 1) o0 = p.scale * tanhf(o0) + p.base;
        emit(row, o0, v1, two);
    }
}

// the forward kernels' EMIT: outputs to p.out [n, n_out]
template <class ARGS>
struct StreamStoreOut {
    const ARGS& p;
    __device__ __forceinline__ void operator()(int row, float o0, float o1, bool two) const {
        p.out[(size_t)row * p.net.n_out] = o0;
        if (two) p.out[(size_t)row * p.net.n_out + 1] = o1;
    }
};

// The rare tile -- the ragged last one, or a caller that saves only one of the pre-activations -- with every condition at run
// time (out of line it cost 900 bytes of scratch per lane for the argument copy: inlined)
template <int H, int CH, class ARGS, class EMIT>
__device__ __forceinline__ void stream_tile_any(const ARGS& p, const StreamLds<H>& lds, int row0, int lane, float b1a, float b1b,
                                                const float (&in3)[3], const EMIT& emit) {
    stream_tile<H, false, CH, -1, -1>(p, lds, row0, lane, b1a, b1b, in3, emit);
}

// Staging of a network into the workgroup's StreamLds (once per workgroup; the caller synchronises)
template <int H, int NW>
__device__ __forceinline__ void stream_stage(const Mlp& net, StreamLds<H>& lds, int tid) {
    constexpr int EIN = 128;
    const int nin = net.S + net.A;
    for (int idx = tid; idx < H * (EIN / 4); idx += NW * 64) {
        const int j = idx / (EIN / 4), q = idx - j * (EIN / 4);
        *reinterpret_cast<float4*>(&lds.w0[j * kStreamLdW + q * 4]) =
            *reinterpret_cast<const float4*>(&net.W0[(size_t)j * EIN + q * 4]);
    }
    for (int idx = tid; idx < EIN * 16; idx += NW * 64) {       // fl[e][lg][ks] = W_first[e][u = 4 ks + lg]
        const int e = idx >> 4, g = (idx >> 2) & 3, ks = idx & 3, u = 4 * ks + g;
        float v = 0.0f;
        if (ks < 3) {
            if (u < net.S) v = net.Ws[e * net.S + u];
            else if (u < nin) v = net.Wa[e * net.A + (u - net.S)];
        }
        lds.fl[idx] = v;
    }
    for (int idx = tid; idx < EIN; idx += NW * 64) lds.bias[idx] = net.A > 0 ? net.bs[idx] + net.ba[idx] : net.bs[idx];
    for (int idx = tid; idx < H; idx += NW * 64) {
        lds.b0[idx] = net.b0[idx];
        lds.w1a[idx] = net.W1[idx];
        lds.w1b[idx] = net.n_out > 1 ? net.W1b[idx] : 0.0f;
    }
}

// FwdArgs is declared by the includer (mlp.hip); the kernel takes the FwdArgs4 of the multi-network launches
template <int H, int NW, class ARGS4>
__device__ __forceinline__ void mlp_forward_stream_body(const ARGS4& p4) {
    __shared__ StreamLds<H> lds;
    const auto& p = p4.net[blockIdx.y];
    const Mlp& net = p.net;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar tile arithmetic)
    stream_stage<H, NW>(net, lds, tid);                         // ---- staging, once per workgroup
    __syncthreads();
    const float b1a = net.b1[0], b1b = net.n_out > 1 ? net.b1b[0] : 0.0f;
    const int tiles = (p.n + kRows - 1) / kRows;
    const int t0 = blockIdx.x * NW + wave, dt = gridDim.x * NW;
    float in3[3] = {0.0f, 0.0f, 0.0f};
    if (t0 < tiles) stream_inputs(p, (long long)t0 * kRows, lane, in3);
    for (int t = t0; t < tiles; t += dt) {
        const int row0 = t * kRows;
        float nxt[3];
        stream_inputs(p, (long long)(t + dt < tiles ? t + dt : t) * kRows, lane, nxt);   // the next tile's inputs land under this tile's MFMAs
        constexpr int CH = NW > 12 ? 2 : 4;
        const StreamStoreOut<typename std::remove_reference<decltype(p)>::type> emit{p};
        if (row0 + kRows <= p.n) {
            const bool save = p.x0_save && p.h1_save, none = !p.x0_save && !p.h1_save, two = net.n_out > 1;
            if (save && !two) stream_tile<H, true, CH, 1, 0>(p, lds, row0, lane, b1a, b1b, in3, emit);          // critics
            else if (none && !two) stream_tile<H, true, CH, 0, 0>(p, lds, row0, lane, b1a, b1b, in3, emit);     // target networks, DDPG actor
            else if (save && two) stream_tile<H, true, CH, 1, 1>(p, lds, row0, lane, b1a, b1b, in3, emit);      // Gaussian actor (policy step)
            else if (none && two) stream_tile<H, true, CH, 0, 1>(p, lds, row0, lane, b1a, b1b, in3, emit);      // Gaussian actor (inference)
            else stream_tile_any<H, CH>(p, lds, row0, lane, b1a, b1b, in3, emit);                               // (one of x0 / h1 saved)
        } else {
            stream_tile_any<H, CH>(p, lds, row0, lane, b1a, b1b, in3, emit);
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) in3[ks] = nxt[ks];
    }
}

}  // namespace rpo_mlp_dev
