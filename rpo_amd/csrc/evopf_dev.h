// EVOPF-v0 device code (reference: rpo/env/electrical_grid/evopf.py): ONE WAVEFRONT PER ENV LANE.
//
// Every lane needs dense linear algebra on its own small matrices -- the 22x22 Newton system of the power-flow
// equation solver (PFFunction, evopf.py:786-855) and the 28x28 block of the equality Jacobian whose inverse defines the
// GRG direction (ineq_partial_grad, evopf.py:596-612).  A thread per lane would keep ~1200 floats of matrix in scratch
// memory; instead the 64 threads of a wave share one lane and THREAD r KEEPS ROW r OF THE MATRIX IN ITS REGISTERS (up to
// 43 floats): it evaluates its own Jacobian row (the column index is a compile-time constant in the unrolled loops, so
// which block of the Jacobian an entry belongs to costs nothing), the pivot search is a 5-step DPP max over (|value|,
// lane) keys, the pivot row is broadcast with v_readlane (lane index in an SGPR), rows are never swapped -- every
// thread just remembers which unknown its row ended up solving -- and only the small result vectors go through LDS.
// History: eliminating in LDS cost 60 us per GRG iteration (dependent LDS round trips); one COLUMN per
// thread 16 us (the serial pivot search and the predicated row swaps ran in every thread); one ROW per thread with a DPP
// pivot search 8.6 us (round 3); round 4: a STATIC elimination order with case14's sparsity compiled in (no search, 154
// instead of 561 row-update pairs; partial pivoting remains as the fallback of a pivot-ratio test) and the whole iteration in
// registers / wave-uniform values ("v2" at the end of this file): 3.4 us -- see tools/evopf_probe.py.  The v1 helpers (flows,
// eq_resid, ineq_resid through LDS; gauss_jordan_rows) still serve the step / residual kernels, PFFunction.backward and the
// fallback.  Workgroup = 4 waves = 4 env lanes, a workspace each: the barriers below only order LDS traffic of one wave.
//
// The network is the IEEE 14-bus case exactly like the reference (case14 hard-wired at evopf.py:211, eq_num = 28 and
// ineq_num = 58 hard-coded at :336-337): the bus classification is compiled in, every number (admittances, limits, costs,
// load / price curves) arrives through the constants buffer the host fills from its case tables (RPO_EVOPF_C_*).
#pragma once
#include "common.h"

namespace rpo_evopf_dev {

constexpr int NB = 14, NG = 5, NE = 5, NY = RPO_EVOPF_ACTION, NS = RPO_EVOPF_STATE, NEQ = 28, NINEQ = 58, NP = 14;
constexpr int PG0 = 0, QG0 = 5, VM0 = 10, VA0 = 24, PE0 = 38;      // blocks of the action vector, evopf.py:278-282
constexpr int NO = 28, NPV = 15, NN = 22;                          // other vars, partial vars (incl. slack angle), Newton
constexpr int T = 24, NAHEAD = 24;
// Battery(...) of evopf.py:243 (per unit)
constexpr float kBLow = 0.1f, kBHigh = 0.8f, kBPmin = -0.2f, kBPmax = 0.2f, kEtaIn = 0.9f, kEtaOut = 0.9f;
constexpr float kBInit = 0.2f;                                      // "empty": low + 0.1, evopf.py:62-63
constexpr float kWe = 5.0f, kWg = 1.0f;                             // evopf.py:244-245
constexpr float kRho = 0.5f, kMinPf = 0.9f, kMaxPf = 1.0f, kRegBias = 0.1f;   // loaders, evopf.py:212,216

// bus classes of case14 (evopf.py:225-235): slack {0}, pv {1,2,5,7}, pq the rest; generator j sits at bus kSpv[j]
__device__ constexpr int kSpv[NG] = {0, 1, 2, 5, 7};
__device__ constexpr int kGenOfBus[NB] = {0, 1, 2, -1, -1, 3, -1, 4, -1, -1, -1, -1, -1, -1};
__device__ constexpr int kLoadSlot[NB] = {-1, 0, 1, 2, 3, 4, -1, -1, 5, 6, 7, 8, 9, 10};   // buses with Pd != 0 (demand.py:46)
// variables the actor sets (evopf.py:287-294; z order = pg at pv gens, vm at gens, [slack angle], pe) ...
__device__ constexpr int kPartialVars[NPV] = {1, 2, 3, 4, 10, 11, 12, 15, 17, 24, 38, 39, 40, 41, 42};
__device__ constexpr int kPartialActions[NP] = {1, 2, 3, 4, 10, 11, 12, 15, 17, 38, 39, 40, 41, 42};
// STATIC ELIMINATION ORDER (round 4).  The equations / unknowns of both linear systems are paired along the network: bus i's
// active-power balance with its angle, and (load buses) its reactive balance with its magnitude -- the diagonal of the
// power-flow Jacobian -- and the buses are taken in the order kBusOrder below.  With the order fixed at compile time (a) no
// pivot search, no data-dependent lane index, no hazard nops; (b) the SPARSITY of the 14-bus network is compiled in: when pivot
// k is taken only the columns in which its row can be non-zero (case14's 20 branches + fill-in, kLive* below, evaluated by
// the compiler from the adjacency masks) are updated -- 154 instead of 561 (broadcast, fma) pairs for the 28 x 43 system of the
// GRG direction, 97 instead of 253 for Newton's 22 x 23.  The order minimises that count (simulated annealing over the 13!
// bus orders, tests/evopf_order.py: generator buses last keeps the J_partial columns sparse longest; minimum degree: 233).
// Accuracy: on 1500 sampled Jacobians (solved states, GRG-like and gross perturbations, Newton's flat start) the static order's
// inverse is as close to the float64 one as partial pivoting's (median 1.5e-7, same p99) whenever min|pivot| / max|pivot| >
// 2^-6; below that (gross perturbations only: near-singular Jacobians) the wave falls back to the partial-pivoting elimination
// (gauss_jordan_rows) -- a wave-uniform branch, see pivots_ok().  RPO_EVOPF_C_FLAGS != RPO_EVOPF_STATIC_OK forces the fallback
// (the default of a table nobody validated; A/B tests).
// The first six "other" variables (slack pg, qg) appear in exactly one equation each with coefficient 1 (P at the slack bus,
// Q at the generator buses): with these rows first the leading 6 x 6 block is the identity and the elimination starts at 6.
__device__ constexpr int kBusOrder[NB - 1] = {2, 7, 11, 10, 13, 4, 12, 9, 8, 6, 3, 1, 5};
__device__ constexpr int kRowOrder[NEQ] = {0, 14, 15, 16, 19, 21, 2, 7, 11, 25, 10, 24, 13, 27, 4, 18, 12, 26, 9, 23, 8, 22,
                                           6, 20, 3, 17, 1, 5};
// ... and the variables the equations determine (evopf.py:290), in elimination order: column k is solved by row k
__device__ constexpr int kOtherVars[NO] = {0, 5, 6, 7, 8, 9, 26, 31, 35, 21, 34, 20, 37, 23, 28, 14, 36, 22, 33, 19, 32, 18,
                                           30, 16, 27, 13, 25, 29};
// Newton system (evopf.py:809-816: P at pv, P at pq, Q at pq  x  vm at pq, va at pv, va at pq), same pairing and order
__device__ constexpr int kKeep[NN] = {2, 7, 11, 25, 10, 24, 13, 27, 4, 18, 12, 26, 9, 23, 8, 22, 6, 20, 3, 17, 1, 5};
__device__ constexpr int kNewtonVars[NN] = {26, 31, 35, 21, 34, 20, 37, 23, 28, 14, 36, 22, 33, 19, 32, 18, 30, 16, 27, 13,
                                            25, 29};
__device__ constexpr int kPvPos[NG - 1] = {20, 0, 21, 1};            // position of P at pv bus j (1, 2, 5, 7) in kKeep
// case14's branches as one adjacency mask per bus (bit k of kAdjMask[i]: Ybus[i][k] != 0, diagonal included); the host
// checks the Ybus it uploads against these masks (EVOPFKernels) -- the reference hard-wires case14 (evopf.py:211)
__device__ constexpr unsigned kAdjMask[NB] = {19, 31, 14, 350, 59, 7216, 456, 192, 9032, 1792, 1568, 6176, 14368, 12544};

// d eq / d var can be non-zero (eq_jac, evopf.py:614-661)
__host__ __device__ constexpr bool jac_struct(int eq, int var) {
    const bool real = eq < NB;
    const int i = real ? eq : eq - NB;
    if (var < QG0) return real && kSpv[var] == i;
    if (var < VM0) return !real && kSpv[var - QG0] == i;
    if (var >= PE0) return real && kSpv[var - PE0] == i;
    return (kAdjMask[i] >> ((var - VM0) % NB)) & 1u;
}

template <int N, int NC>
struct LiveTab { bool v[N][NC]; };     // v[k][c]: column c (> k) of pivot row k can be non-zero when pivot k is taken

// Symbolic Gauss-Jordan of an N x NC pattern in the fixed order (pivot k = row k, column k), starting at pivot K0
template <int N, int NC, int K0, typename F>
__host__ __device__ constexpr LiveTab<N, NC> symbolic_gj(F pattern) {
    LiveTab<N, NC> t{};
    bool p[N][NC] = {};
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < NC; ++c) p[r][c] = pattern(r, c);
    for (int k = K0; k < N; ++k) {
        for (int c = k + 1; c < NC; ++c) t.v[k][c] = p[k][c];
        for (int r = 0; r < N; ++r)
            if (r != k && p[r][k])
                for (int c = k + 1; c < NC; ++c) p[r][c] = p[r][c] || p[k][c];
    }
    return t;
}
struct PatGrg {       // [J_other | J_partial], rows kRowOrder
    __host__ __device__ constexpr bool operator()(int r, int c) const {
        return jac_struct(kRowOrder[r], c < NO ? kOtherVars[c] : kPartialVars[c - NO]);
    }
};
struct PatNewton {    // [J_newton | g]
    __host__ __device__ constexpr bool operator()(int r, int c) const { return c == NN || jac_struct(kKeep[r], kNewtonVars[c]); }
};
struct PatNewtonT {   // [J_newton^T | rhs] (PFFunction.backward)
    __host__ __device__ constexpr bool operator()(int r, int c) const { return c == NN || jac_struct(kKeep[c], kNewtonVars[r]); }
};
__device__ constexpr LiveTab<NN, NN + 1> kLiveNewton = symbolic_gj<NN, NN + 1, 0>(PatNewton{});
__device__ constexpr LiveTab<NN, NN + 1> kLiveNewtonT = symbolic_gj<NN, NN + 1, 0>(PatNewtonT{});
__device__ constexpr LiveTab<NEQ, NY> kLiveGrg0 = symbolic_gj<NEQ, NY, 0>(PatGrg{});   // incl. the six unit pivots (v2's extra row)
struct TabGrg0 { static __device__ constexpr bool live(int k, int c) { return kLiveGrg0.v[k][c]; } };
struct TabNewton { static __device__ constexpr bool live(int k, int c) { return kLiveNewton.v[k][c]; } };
struct TabNewtonT { static __device__ constexpr bool live(int k, int c) { return kLiveNewtonT.v[k][c]; } };

struct Ws {                       // per-wave workspace in LDS
    float c[RPO_EVOPF_CONSTS_LEN];
    float D[NO][NPV + 1];         // inv(J_other) J_partial = -dynz_dz (evopf.py:598)
    float s[NS + 3];
    float a[NY + 1];
    float cs[NB], sn[NB], vr[NB], vi[NB], t1[NB], t2[NB];
    float eq[NEQ];
    float ineq[NINEQ + 2];
    float dir[NY + 1];            // direct gradient / step
    alignas(16) float fp[NPV + 1];   // full_partial_grad
    float old[NY + 1];            // momentum term of grad_steps
    alignas(16) float vec[64];    // scratch
};

// One wavefront per Ws (the kernels index their workspace by wave): the LDS instructions of a wave
// execute in issue order, so a cross-lane hand-over through LDS needs neither s_barrier nor the s_waitcnt 0 that
// __syncthreads() implies -- only that the compiler keeps stores and loads in program order (act_project 133.8 -> 131.5 us).
// Lane of the wave (a workgroup may hold several waves, each solving its own env lane with its own Ws)
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (RPO_WAVE - 1)); }

__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void load_consts(Ws& w, const float* __restrict__ consts) {
    for (int i = lane_id(); i < RPO_EVOPF_CONSTS_LEN; i += RPO_WAVE) w.c[i] = consts[i];
}
__device__ __forceinline__ float Yr(const Ws& w, int i, int k) { return w.c[RPO_EVOPF_C_YR + i * NB + k]; }
__device__ __forceinline__ float Yi(const Ws& w, int i, int k) { return w.c[RPO_EVOPF_C_YI + i * NB + k]; }

// Charge-rate box shrunk by the state of charge (Battery.update_bound / ineq_resid, evopf.py:110-116,135-141)
__device__ __forceinline__ void battery_bounds(float soc, float& p_max, float& p_min) {
    RPO_FP_STRICT
    p_max = fminf(kBPmax, kBHigh - soc) / kEtaIn;
    p_min = kEtaOut * fmaxf(kBPmin, kBLow - soc);
}

// Box of basic action j of the lane whose observation is in w.s (EVOPFEnv.update, evopf.py:769-783)
__device__ __forceinline__ void partial_box(const Ws& w, int j, float& lo, float& hi) {
    if (j < 4) { lo = w.c[RPO_EVOPF_C_PMIN + 1 + j]; hi = w.c[RPO_EVOPF_C_PMAX + 1 + j]; }
    else if (j < 9) { lo = w.c[RPO_EVOPF_C_VMIN + kSpv[j - 4]]; hi = w.c[RPO_EVOPF_C_VMAX + kSpv[j - 4]]; }
    else battery_bounds(w.s[2 * NB + j - 9], hi, lo);
}

// Ybus operands a thread needs again in every Newton / GRG iteration, kept in registers (they are constants; read from LDS they
// were two reads per Jacobian entry and four per term of the admittance products, behind LDS stores the compiler cannot move
// them across): column `tid` for flows(), row `i` for the Jacobian row of the equation a thread owns.  Same values, same
// operations.
struct YVec {
    float yr[NB], yi[NB];
};
__device__ __forceinline__ YVec y_column(const Ws& w, int k) {
    YVec y;
#pragma unroll
    for (int i = 0; i < NB; ++i) { y.yr[i] = Yr(w, i, k); y.yi[i] = Yi(w, i, k); }
    return y;
}
__device__ __forceinline__ YVec y_row(const Ws& w, int i) {
    YVec y;
#pragma unroll
    for (int k = 0; k < NB; ++k) { y.yr[k] = Yr(w, i, k); y.yi[k] = Yi(w, i, k); }
    return y;
}

// flows() with the thread's Ybus column in registers (y_column(w, tid < NB ? tid : 0))
__device__ __forceinline__ void flows(Ws& w, const YVec& yc) {
    const int tid = lane_id();
    if (tid < NB) {
        float sn, cs;
        sincosf(w.a[VA0 + tid], &sn, &cs);
        const float vm = w.a[VM0 + tid];
        w.cs[tid] = cs; w.sn[tid] = sn; w.vr[tid] = vm * cs; w.vi[tid] = vm * sn;
    }
    sync();
    if (tid < NB) {
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int i = 0; i < NB; ++i) {                       // row vector @ matrix, as written at :527-528
            t1 += w.vr[i] * yc.yr[i] - w.vi[i] * yc.yi[i];
            t2 += w.vr[i] * yc.yi[i] + w.vi[i] * yc.yr[i];
        }
        w.t1[tid] = t1; w.t2[tid] = t2;
    }
    sync();
}

// cos / sin / rectangular voltages and the two admittance products of eq_resid / eq_jac (evopf.py:523-528,623-630)
__device__ __forceinline__ void flows(Ws& w) {
    const int tid = lane_id();
    if (tid < NB) {
        float sn, cs;
        sincosf(w.a[VA0 + tid], &sn, &cs);
        const float vm = w.a[VM0 + tid];
        w.cs[tid] = cs; w.sn[tid] = sn; w.vr[tid] = vm * cs; w.vi[tid] = vm * sn;
    }
    sync();
    if (tid < NB) {
        float t1 = 0.0f, t2 = 0.0f;
        for (int i = 0; i < NB; ++i) {                       // row vector @ matrix, as written at :527-528
            t1 += w.vr[i] * Yr(w, i, tid) - w.vi[i] * Yi(w, i, tid);
            t2 += w.vr[i] * Yi(w, i, tid) + w.vi[i] * Yr(w, i, tid);
        }
        w.t1[tid] = t1; w.t2[tid] = t2;
    }
    sync();
}

// eq_resid (evopf.py:520-546) of (w.s, w.a) into w.eq; flows() must be current
__device__ __forceinline__ void eq_resid(Ws& w) {
    const int tid = lane_id();
    if (tid < NEQ) {
        const int i = tid < NB ? tid : tid - NB;
        const int g = kGenOfBus[i];
        if (tid < NB) {
            const float inj = g >= 0 ? w.a[PG0 + g] + w.a[PE0 + g] : 0.0f;
            w.eq[tid] = (inj - w.s[i]) - (w.vr[i] * w.t1[i] + w.vi[i] * w.t2[i]);
        } else {
            const float inj = g >= 0 ? w.a[QG0 + g] : 0.0f;
            w.eq[tid] = (inj - w.s[NB + i]) - (-w.vr[i] * w.t2[i] + w.vi[i] * w.t1[i]);
        }
    }
    sync();
}

// ineq_resid (evopf.py:548-563) into w.ineq (signed)
__device__ __forceinline__ void ineq_resid(Ws& w) {
    RPO_FP_STRICT
    const int tid = lane_id();
    if (tid < NINEQ) {
        float r;
        if (tid < 5) r = w.a[PG0 + tid] - w.c[RPO_EVOPF_C_PMAX + tid];
        else if (tid < 10) r = w.c[RPO_EVOPF_C_PMIN + tid - 5] - w.a[PG0 + tid - 5];
        else if (tid < 15) r = w.a[QG0 + tid - 10] - w.c[RPO_EVOPF_C_QMAX + tid - 10];
        else if (tid < 20) r = w.c[RPO_EVOPF_C_QMIN + tid - 15] - w.a[QG0 + tid - 15];
        else if (tid < 34) r = w.a[VM0 + tid - 20] - w.c[RPO_EVOPF_C_VMAX + tid - 20];
        else if (tid < 48) r = w.c[RPO_EVOPF_C_VMIN + tid - 34] - w.a[VM0 + tid - 34];
        else {
            const int j = (tid - 48) % NE;
            float p_max, p_min;
            battery_bounds(w.s[2 * NB + j], p_max, p_min);
            r = tid < 53 ? w.a[PE0 + j] - p_max : p_min - w.a[PE0 + j];
        }
        w.ineq[tid] = r;
    }
    sync();
}

// One entry of eq_jac (evopf.py:614-661): d eq[row] / d action[var], in the reference's orientation and with its sign
// for the battery columns (d real / d pe = -I at :639-640 although eq_resid adds +pe: reproduced, DESIGN.md hazard E1).
__device__ __forceinline__ float jac_entry(const Ws& w, int row, int var) {
    const bool real = row < NB;
    const int i = real ? row : row - NB;
    if (var < VM0) {                                           // pg / qg selectors
        const bool mine = real ? var < QG0 : var >= QG0;
        const int g = var < QG0 ? var : var - QG0;
        return (mine && kSpv[g] == i) ? 1.0f : 0.0f;
    }
    if (var >= PE0) return (real && kSpv[var - PE0] == i) ? -1.0f : 0.0f;
    const bool dvm = var < VA0;
    const int k = dvm ? var - VM0 : var - VA0;
    const float yr = Yr(w, i, k), yi = Yi(w, i, k);
    const float p = dvm ? w.cs[k] : -w.vi[k];                  // d vr_k / d var
    const float q = dvm ? w.sn[k] : w.vr[k];                   // d vi_k / d var
    const float pi_ = dvm ? w.cs[i] : -w.vi[i], qi_ = dvm ? w.sn[i] : w.vr[i];
    const float d = (i == k) ? 1.0f : 0.0f;
    if (real)
        return -(d * pi_ * w.t1[i]) - w.vr[i] * (yr * p - yi * q) - (d * qi_ * w.t2[i]) - w.vi[i] * (yi * p + yr * q);
    return (d * pi_ * w.t2[i]) + w.vr[i] * (yi * p + yr * q) - (d * qi_ * w.t1[i]) - w.vi[i] * (yr * p - yi * q);
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {          // lane: wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// max over lanes 0..31 of a key, returned to every thread (DPP row shifts, then row 0's result is merged into row 1)
__device__ __forceinline__ unsigned half_wave_umax(unsigned key) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, false); key = key > o ? key : o;   // row_shr:1
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x112, 0xf, 0xf, false); key = key > o ? key : o;   // row_shr:2
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x114, 0xf, 0xf, false); key = key > o ? key : o;   // row_shr:4
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x118, 0xf, 0xf, false); key = key > o ? key : o;   // row_shr:8
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x142, 0xa, 0xf, false); key = key > o ? key : o;   // row_bcast:15
    return (unsigned)__builtin_amdgcn_readlane((int)key, 31);
}

// Per-thread view of the equation (Jacobian row) a thread owns: everything of eq_jac (evopf.py:614-661) that does not
// depend on the column.  With A = Yr[i][k] p_k - Yi[i][k] q_k and B = Yi[i][k] p_k + Yr[i][k] q_k, where (p, q) =
// d(vr_k, vi_k)/d var = (cos, sin) for vm_k and (-vi, vr) for va_k:
//   real row      d/dvar = -(vr_i A + vi_i B) - [i == k] (p_i t1_i + q_i t2_i)
//   reactive row  d/dvar =  (vr_i B - vi_i A) + [i == k] (p_i t2_i - q_i t1_i)
struct RowCoef {
    int i;            // bus of the equation
    bool real;        // active (true) or reactive power balance
    float ca, cb;     // coefficients of A and B
    float dg_vm, dg_va;   // diagonal terms for the vm / va column of the same bus
};

__device__ __forceinline__ RowCoef row_coef(const Ws& w, int eq) {
    RowCoef c;
    c.real = eq < NB;
    c.i = c.real ? eq : eq - NB;
    const float vr = w.vr[c.i], vi = w.vi[c.i], cs = w.cs[c.i], sn = w.sn[c.i], t1 = w.t1[c.i], t2 = w.t2[c.i];
    c.ca = c.real ? -vr : -vi;
    c.cb = c.real ? -vi : vr;
    c.dg_vm = c.real ? -(cs * t1 + sn * t2) : (cs * t2 - sn * t1);
    c.dg_va = c.real ? -(-vi * t1 + vr * t2) : (-vi * t2 - vr * t1);
    return c;
}

// eq_jac entry (this thread's equation, variable `var`); `var` is a compile-time constant at every call site, so only
// the block it names survives.  Battery columns carry the reference's sign (hazard E1).
__device__ __forceinline__ float jac_row_entry(const Ws& w, const RowCoef& c, int var, const YVec& yrow) {
    if (var < QG0) return (c.real && kSpv[var] == c.i) ? 1.0f : 0.0f;
    if (var < VM0) return (!c.real && kSpv[var - QG0] == c.i) ? 1.0f : 0.0f;
    if (var >= PE0) return (c.real && kSpv[var - PE0] == c.i) ? -1.0f : 0.0f;
    const bool dvm = var < VA0;
    const int k = dvm ? var - VM0 : var - VA0;
    const float yr = yrow.yr[k], yi = yrow.yi[k];              // (k is a compile-time constant at every call site)
    const float p = dvm ? w.cs[k] : -w.vi[k], q = dvm ? w.sn[k] : w.vr[k];
    const float A = yr * p - yi * q, B = yi * p + yr * q;
    return c.ca * A + c.cb * B + (c.i == k ? (dvm ? c.dg_vm : c.dg_va) : 0.0f);
}
__device__ __forceinline__ float jac_row_entry(const Ws& w, const RowCoef& c, int var) {
    if (var < QG0) return (c.real && kSpv[var] == c.i) ? 1.0f : 0.0f;
    if (var < VM0) return (!c.real && kSpv[var - QG0] == c.i) ? 1.0f : 0.0f;
    if (var >= PE0) return (c.real && kSpv[var - PE0] == c.i) ? -1.0f : 0.0f;
    const bool dvm = var < VA0;
    const int k = dvm ? var - VM0 : var - VA0;
    const float yr = Yr(w, c.i, k), yi = Yi(w, c.i, k);
    const float p = dvm ? w.cs[k] : -w.vi[k], q = dvm ? w.sn[k] : w.vr[k];
    const float A = yr * p - yi * q, B = yi * p + yr * q;
    return c.ca * A + c.cb * B + (c.i == k ? (dvm ? c.dg_vm : c.dg_va) : 0.0f);
}

// Gauss-Jordan with partial pivoting on an N x NC system distributed one ROW per thread (threads >= N pass zeros).
// Rows are not swapped and pivot rows are not normalised: on return thread r's row solves unknown `mycol` and
// row[c] / mypiv (c >= N) is entry [mycol][c] of inv(A) @ (the trailing columns).  The first K0 rows must be the unit
// rows e_0 .. e_{K0-1} within the first K0 columns, and the other rows zero there.
template <int N, int NC, int K0>
__device__ __forceinline__ void gauss_jordan_rows(float (&row)[NC], int& mycol, float& mypiv) {
    const int lane = lane_id();
    bool used = lane < K0 || lane >= N;
    mycol = lane < K0 ? lane : 0;
    mypiv = 1.0f;
#pragma unroll
    for (int k = K0; k < N; ++k) {
        const unsigned key = used ? 0u : ((__float_as_uint(fabsf(row[k])) & ~31u) | (unsigned)lane);
        const int p = (int)(half_wave_umax(key) & 31u);
        const float piv = lane_bcast(row[k], p);
        const bool is_p = lane == p;
        const float f = is_p ? 0.0f : -row[k] * (1.0f / piv);
        if (is_p) { used = true; mycol = k; mypiv = piv; }
        // two columns per instruction (v_pk_fma_f32: the same IEEE fma per component; the pivot row's pair sits in an SGPR pair).
        // Pairs are (even, odd) columns whatever k is: 64-bit register operands must be even-aligned, a pairing that moved with
        // k made the compiler re-pack the row with v_pk_mov_b32 at every pivot.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 f2 = {f, f};
        const int start = (k + 2) & ~1;                              // first even column behind the pivot column
        if ((k + 1) & 1) row[k + 1] = fmaf(f, lane_bcast(row[k + 1], p), row[k + 1]);
#pragma unroll
        for (int c = start; c + 1 < NC; c += 2) {
            const f32x2 pv = {lane_bcast(row[c], p), lane_bcast(row[c + 1], p)};
            f32x2 rv = {row[c], row[c + 1]};
            rv = __builtin_elementwise_fma(f2, pv, rv);
            row[c] = rv.x; row[c + 1] = rv.y;
        }
        if ((NC & 1) && start < NC) row[NC - 1] = fmaf(f, lane_bcast(row[NC - 1], p), row[NC - 1]);
    }
}

// Gauss-Jordan in the STATIC order: pivot k is row k (lane k), column k; only the columns Tab::live(k, c) marks are
// touched.  Rows are not normalised: on return thread r's row solves unknown r and row[c] / pivot (c >= N) is entry [r][c]
// of inv(A) @ (the trailing columns); the return value is the thread's pivot.  Same K0 convention as gauss_jordan_rows.
template <int N, int NC, int K0, typename Tab>
__device__ __forceinline__ float gauss_jordan_static(float (&row)[NC]) {
    const int lane = lane_id();
    float mypiv = 1.0f;
#pragma unroll
    for (int k = K0; k < N; ++k) {
        const float piv = lane_bcast(row[k], k);                     // v_readlane with an immediate lane
        const float f = lane == k ? 0.0f : -row[k] * (1.0f / piv);
        mypiv = lane == k ? piv : mypiv;
#pragma unroll
        for (int c = k + 1; c < NC; ++c)
            if (Tab::live(k, c)) row[c] = fmaf(f, lane_bcast(row[c], k), row[c]);
    }
    return mypiv;
}

// The static order's acceptance test (wave-uniform): every pivot within 2^-6 of the largest one.  NaN / zero / infinite
// pivots fail it.  Lanes [K0, N) hold the pivots.
template <int N, int K0>
__device__ __forceinline__ bool pivots_ok(float mypiv) {
    const int lane = lane_id();
    const bool mine = lane >= K0 && lane < N;
    const float a = fabsf(mypiv);
    const float amax = rpo_wave_max_nonneg(mine ? a : 0.0f);          // (NaN pivots: a > ... below is false)
    const bool fine = !mine || (a > amax * 0.015625f && a < 3.0e38f);
    return __builtin_amdgcn_ballot_w64(fine) == ~0ull;
}

// =====================================================================================================================
// v2 of the equation solver and the GRG projection (round 4): the whole iteration in registers and wave-uniform values.
//
// v1 (rounds 1-3; its solver functions were removed in round 4, its helpers above remain) handed every intermediate over through LDS (cos / sin / rectangular voltages -> admittance products -> residuals
// -> Jacobian rows -> D -> two products with D), each a store, a wave barrier and dependent loads: with ONE resident wave
// nothing hides those round trips.  Here:
//   * lane r < 28 owns equation kRowOrder[r] (both systems: Newton's 22 x 22 is the block of rows / columns 6..27), lanes
//     32..45 own bus lane - 32; every lane takes cos / sin of the angle of ITS bus, so the 4 x 14 per-bus values every row
//     needs are wave-uniform v_readlane results (scalar operands of the FMAs), not LDS traffic;
//   * a row computes its own admittance products t1_i, t2_i (14 terms from the uniforms and its Ybus column), its own
//     residual and its own Jacobian row: entry (row i, bus k) = p_k alpha_k + q_k beta_k (+ the diagonal term), alpha / beta
//     formed once per bus (the same for the |v| and the angle column);
//   * the reduced gradient needs g_p - D^T g_o and then -D (that): the first is obtained by one more ROW that takes part in
//     the elimination -- lane 28 holds [g_o^T | g_p^T]; eliminating its first 28 entries with the pivot rows leaves
//     g_p - g_o^T inv(J_o) J_p in its trailing 15 -- at no cost (one more lane of the same instructions; the six unit pivots
//     of the slack / qg equations are applied to it first); the second is 15 FMAs per row on its own registers;
//   * `a` stays in LDS (read by bus, updated by the owner of each component); nothing else does.
// Same arithmetic as the reference (evopf.py:520-546,596-661,786-855; rpo_ddpg.py:266-305), other association of the
// Jacobian entries' products than v1 (float32 round-off).
#ifndef RPO_EVOPF_RCP
#define RPO_EVOPF_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
struct Volt { float cs, sn, vr, vi; };

struct RowLane {
    bool is_row, real, is_extra;
    int bus;                 // bus of the equation (row lanes), lane - 32 (bus lanes 32..45), else 0
    int ia, ib, dem;         // w.a indices of the injections of the row's residual (NY: none -> w.a[NY] == 0), w.s index of its demand
    float dlt[NB];           // [bus == k] for row lanes, else 0
    YVec yrow, ycol;         // Ybus row (Jacobian) and column (admittance products) of the bus; zero for non-row lanes
    float hi, lo;            // bounds of action component `lane` (lanes < 43): ineq_resid (evopf.py:548-563) per component
    int colpos;              // position of action component `lane` in the [other | partial] column order (NY: none)
    int own_var;             // row lanes: the unknown of pivot `lane` (kOtherVars[lane]); lanes 32..46: kPartialVars[lane - 32]; else NY
};

__device__ __forceinline__ RowLane make_row_lane(Ws& w) {
    RPO_FP_STRICT
    RowLane L;
    const int lane = lane_id();
    L.is_row = lane < NEQ;
    L.is_extra = lane == NEQ;
    int eq = 0;
#pragma unroll
    for (int r = 0; r < NEQ; ++r) eq = lane == r ? kRowOrder[r] : eq;
    L.real = eq < NB;
    const int bl = lane - 32;
    L.bus = L.is_row ? (L.real ? eq : eq - NB) : ((bl >= 0 && bl < NB) ? bl : 0);
    int g = -1;
#pragma unroll
    for (int j = 0; j < NG; ++j) g = kSpv[j] == L.bus ? j : g;
    L.ia = (L.is_row && g >= 0) ? (L.real ? PG0 + g : QG0 + g) : NY;
    L.ib = (L.is_row && L.real && g >= 0) ? PE0 + g : NY;
    L.dem = L.real ? L.bus : NB + L.bus;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        L.dlt[k] = (L.is_row && L.bus == k) ? 1.0f : 0.0f;
        L.yrow.yr[k] = L.is_row ? Yr(w, L.bus, k) : 0.0f;
        L.yrow.yi[k] = L.is_row ? Yi(w, L.bus, k) : 0.0f;
        L.ycol.yr[k] = L.is_row ? Yr(w, k, L.bus) : 0.0f;
        L.ycol.yi[k] = L.is_row ? Yi(w, k, L.bus) : 0.0f;
    }
    // per-component bounds (the 58 inequalities are an upper and a lower bound on pg, qg, |v|, pe; angles are free)
    L.hi = 3.0e38f; L.lo = -3.0e38f;
    if (lane < QG0) { L.hi = w.c[RPO_EVOPF_C_PMAX + lane]; L.lo = w.c[RPO_EVOPF_C_PMIN + lane]; }
    else if (lane < VM0) { L.hi = w.c[RPO_EVOPF_C_QMAX + lane - QG0]; L.lo = w.c[RPO_EVOPF_C_QMIN + lane - QG0]; }
    else if (lane < VA0) { L.hi = w.c[RPO_EVOPF_C_VMAX + lane - VM0]; L.lo = w.c[RPO_EVOPF_C_VMIN + lane - VM0]; }
    else if (lane >= PE0 && lane < NY) battery_bounds(w.s[2 * NB + lane - PE0], L.hi, L.lo);
    if (lane == 0) w.a[NY] = 0.0f;
    L.colpos = NY;
#pragma unroll
    for (int c = 0; c < NY; ++c) L.colpos = (c < NO ? kOtherVars[c] : kPartialVars[c - NO]) == lane ? c : L.colpos;
    L.own_var = NY;
#pragma unroll
    for (int r = 0; r < NO; ++r) L.own_var = lane == r ? kOtherVars[r] : L.own_var;
#pragma unroll
    for (int p = 0; p < NPV; ++p) L.own_var = lane == 32 + p ? kPartialVars[p] : L.own_var;
    return L;
}

// cos / sin / rectangular voltage of the lane's bus from w.a, and the 4 x 14 wave-uniform copies (bus lanes 32 + k)
struct Uniforms { float cs[NB], sn[NB], vr[NB], vi[NB]; };
__device__ __forceinline__ Volt own_volt(const Ws& w, const RowLane& L) {
    Volt v;
    sincosf(w.a[VA0 + L.bus], &v.sn, &v.cs);
    const float vm = w.a[VM0 + L.bus];
    v.vr = vm * v.cs; v.vi = vm * v.sn;
    return v;
}
__device__ __forceinline__ Uniforms bus_uniforms(const Volt& v) {
    Uniforms u;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        u.cs[k] = lane_bcast(v.cs, 32 + k); u.sn[k] = lane_bcast(v.sn, 32 + k);
        u.vr[k] = lane_bcast(v.vr, 32 + k); u.vi[k] = lane_bcast(v.vi, 32 + k);
    }
    return u;
}

// The row's admittance products (evopf.py:527-528, the sums in v1's order) and its residual (:520-546)
__device__ __forceinline__ float row_residual(const Ws& w, const RowLane& L, const Volt& v, const Uniforms& u, float& t1, float& t2) {
    t1 = 0.0f; t2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        t1 += u.vr[k] * L.ycol.yr[k] - u.vi[k] * L.ycol.yi[k];
        t2 += u.vr[k] * L.ycol.yi[k] + u.vi[k] * L.ycol.yr[k];
    }
    const float inj = L.real ? w.a[L.ia] + w.a[L.ib] : w.a[L.ia];
    const float flow = L.real ? (v.vr * t1 + v.vi * t2) : (-v.vr * t2 + v.vi * t1);
    return L.is_row ? (inj - w.s[L.dem]) - flow : 0.0f;
}

// Columns [C0, C0 + NC) (in the [other | partial] order) of the row's Jacobian row (eq_jac, evopf.py:614-661; battery
// columns with the reference's sign, hazard E1) into out[0 .. NC)
template <int C0, int NC, int NOUT>
__device__ __forceinline__ void jacobian_row(const RowLane& L, const Volt& v, const Uniforms& u, float t1, float t2, float (&out)[NOUT]) {
    const float ca = L.is_row ? (L.real ? -v.vr : -v.vi) : 0.0f, cb = L.is_row ? (L.real ? -v.vi : v.vr) : 0.0f;
    // (the diagonal terms enter every column as dlt[k] * dg with dlt = [bus == k]: clamped to finite values so that a non-finite
    //  term of a diverged Newton iterate stays on the diagonal -- 0 * inf would put a NaN into every entry of the row)
    const float dg_vm = fminf(fmaxf(L.real ? -(v.cs * t1 + v.sn * t2) : (v.cs * t2 - v.sn * t1), -3.0e38f), 3.0e38f);
    const float dg_va = fminf(fmaxf(L.real ? -(-v.vi * t1 + v.vr * t2) : (-v.vi * t2 - v.vr * t1), -3.0e38f), 3.0e38f);
    float alpha[NB], beta[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        alpha[k] = ca * L.yrow.yr[k] + cb * L.yrow.yi[k];
        beta[k] = cb * L.yrow.yr[k] - ca * L.yrow.yi[k];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = C0 + c, var = col < NO ? kOtherVars[col] : kPartialVars[col - NO];
        float e;
        if (var < QG0) e = (L.is_row && L.real && kSpv[var] == L.bus) ? 1.0f : 0.0f;
        else if (var < VM0) e = (L.is_row && !L.real && kSpv[var - QG0] == L.bus) ? 1.0f : 0.0f;
        else if (var >= PE0) e = (L.is_row && L.real && kSpv[var - PE0] == L.bus) ? -1.0f : 0.0f;
        else {
            const bool dvm = var < VA0;
            const int k = dvm ? var - VM0 : var - VA0;
            const float p = dvm ? u.cs[k] : -u.vi[k], q = dvm ? u.sn[k] : u.vr[k];
            e = fmaf(L.dlt[k], dvm ? dg_vm : dg_va, fmaf(p, alpha[k], q * beta[k]));
        }
        out[c] = e;
    }
}

// Static-order Gauss-Jordan on columns [OFF, OFF + NC) of the 28-row system held one row per lane (array index = column -
// OFF): pivots KB .. KE-1, pivot k = lane k; pivots below KUNIT are exactly 1 (the slack / qg equations).  Returns the
// lane's pivot (1 for lanes that own none).
template <int KB, int KE, int KUNIT, int OFF, int NC, typename Tab>
__device__ __forceinline__ float gj_static(float (&row)[NC]) {
    const int lane = lane_id();
    float mypiv = 1.0f;
#pragma unroll
    for (int k = KB; k < KE; ++k) {
        const int j = k - OFF;
        float f;
        if (k < KUNIT) {
            // unit pivots (slack pg / qg equations): only the extra row (lane NEQ: [g_o | g_p]) has an entry in their columns --
            // nothing to do when the bound of that variable is not violated (wave-uniform skip of the whole step)
            if (lane_bcast(row[j], NEQ) == 0.0f) continue;
            f = lane == k ? 0.0f : -row[j];
        } else {
            const float piv = lane_bcast(row[j], k);
            // multiplier with the hardware reciprocal (1 ulp): an error of the multiplier only leaves a residue of that relative
            // size in the eliminated entry -- the size of the elimination's own rounding errors -- and takes the twelve
            // dependent instructions of a correctly rounded division off each of the 22 links of the chain
            f = lane == k ? 0.0f : -row[j] * RPO_EVOPF_RCP(piv);
            mypiv = lane == k ? piv : mypiv;
        }
#pragma unroll
        for (int c = j + 1; c < NC; ++c)
            if (Tab::live(j, c)) row[c] = fmaf(f, lane_bcast(row[c], k), row[c]);
    }
    return mypiv;
}

// The same pivots by partial pivoting (the fallback of the static order): rows are not swapped, `mycol` = the pivot a
// lane ended up owning (its own lane index where it owns none); lanes outside [KB, KE) are never chosen but are updated.
template <int KB, int KE, int OFF, int NC>
__device__ __forceinline__ float gj_dynamic(float (&row)[NC], int& mycol) {
    const int lane = lane_id();
    bool used = lane < KB || lane >= KE;
    mycol = lane;
    float mypiv = 1.0f;
#pragma unroll
    for (int k = KB; k < KE; ++k) {
        const int j = k - OFF;
        const unsigned key = used ? 0u : ((__float_as_uint(fabsf(row[j])) & ~31u) | (unsigned)lane);
        const int p = (int)(half_wave_umax(key) & 31u);
        const float piv = lane_bcast(row[j], p);
        const bool is_p = lane == p;
        const float f = is_p ? 0.0f : -row[j] * (1.0f / piv);
        if (is_p) { used = true; mycol = k; mypiv = piv; }
#pragma unroll
        for (int c = j + 1; c < NC; ++c) row[c] = fmaf(f, lane_bcast(row[c], p), row[c]);
    }
    return mypiv;
}

// complete_partial v2 (PFFunction.forward, evopf.py:789-855): Newton on rows / columns 6..27 with the lane's own stop test,
// then qg and the slack generation from rows 0..5.  zj: basic action z[lane] in lanes 0..13.
__device__ __forceinline__ int complete_partial_v2(Ws& w, const RowLane& L, float zj, float tol, int max_iters) {
    const int tid = lane_id();
    if (tid < NY) {
        float v = 0.0f;                                        // qg and the slack pg start at zero (:806-807)
        if (tid >= VM0 && tid < VA0) v = w.c[RPO_EVOPF_C_VM_INIT + tid - VM0];       // load-bus guesses (:802)
        else if (tid >= VA0 && tid < PE0) v = w.c[RPO_EVOPF_C_VA_INIT + tid - VA0];  // (:798,803-805)
        w.a[tid] = v;
    }
    sync();
    if (tid < NP) w.a[kPartialActions[tid]] = zj;              // (:796-799)
    sync();
    const bool force_dyn = w.c[RPO_EVOPF_C_FLAGS] != RPO_EVOPF_STATIC_OK;
    int it = 0;
    for (; it < max_iters;) {
        const Volt v = own_volt(w, L);
        const Uniforms u = bus_uniforms(v);
        float t1, t2;
        const float resid = row_residual(w, L, v, u, t1, t2);
        float delta = 0.0f;
        int unknown = L.own_var;                               // w.a index of the unknown this lane's pivot solves
        bool solved = false;
        if (!force_dyn) {
            float row[NN + 1];                                 // columns 6..27 of the row | g
            jacobian_row<6, NN>(L, v, u, t1, t2, row);
            row[NN] = resid;
            const float mypiv = gj_static<6, NEQ, 6, 6, NN + 1, TabNewton>(row);
            solved = pivots_ok<NEQ, 6>(mypiv);
            delta = row[NN] / mypiv;                           // delta = inv(J) g (:832), unknown `tid`
        }
        if (!solved) {                                         // partial pivoting (near-singular Jacobians, or forced)
            float row[NN + 1];
            jacobian_row<6, NN>(L, v, u, t1, t2, row);
            row[NN] = resid;
            int mycol;
            const float mypiv = gj_dynamic<6, NEQ, 6, NN + 1>(row, mycol);
            delta = row[NN] / mypiv;
#pragma unroll
            for (int r = 0; r < NO; ++r) unknown = mycol == r ? kOtherVars[r] : unknown;
        }
        const bool mine = tid >= 6 && tid < NEQ;
        if (mine) w.a[unknown] -= delta;
        sync();
        ++it;
        if (sqrtf(rpo_wave_sum_rows(mine ? delta * delta : 0.0f)) < tol) break;   // torch.norm(delta) < tol (:834), this lane only
    }
    {   // qg at the generators and the slack generation from the remaining equations (:844-848), with qg = pg_slack = 0
        const Volt v = own_volt(w, L);
        const Uniforms u = bus_uniforms(v);
        float t1, t2;
        const float resid = row_residual(w, L, v, u, t1, t2);
        if (tid < 6) w.a[L.own_var] = -resid;
        sync();
    }
    return it;
}

// One evaluation of the reduced gradient at w.a (ineq_partial_grad, evopf.py:590-612) and -- when `lr_step` -- the GRG step
// a -= lr * grad + momentum * old (rpo_ddpg.py:279-287).  `first`: skip the stop test (rpo_ddpg.py:271, step == 0).
// Returns false when the stop test ended the loop (nothing was changed).  With `dir_out` the gradient is also written
// there ([43], by component).
__device__ __forceinline__ bool grg_iteration_v2(Ws& w, const RowLane& L, bool first, float corr_eps, float lr, float momentum,
                                                 bool apply, float* dir_out) {
    const int tid = lane_id();
    const Volt v = own_volt(w, L);
    const Uniforms u = bus_uniforms(v);
    float t1, t2;
    const float resid = row_residual(w, L, v, u, t1, t2);
    // per-component inequality residuals (:548-563) and ineq_grad_new (:590-594): +-1 per violated bound
    const float av = w.a[tid < NY ? tid : NY];
    const float up = av - L.hi, dn = L.lo - av;
    if (!first) {
        float m = fabsf(resid);                                // (0 outside the row lanes)
        if (tid < NY) m = fmaxf(m, fmaxf(up, dn));
        if (!(rpo_wave_max_nonneg(fmaxf(m, 0.0f)) > corr_eps)) return false;
    }
    const float g = tid < NY ? ((up > 0.0f ? 1.0f : 0.0f) - (dn > 0.0f ? 1.0f : 0.0f)) : 0.0f;
    w.vec[L.colpos] = g;                                       // (lanes >= NY park a zero in slot NY: never read)
    sync();
    const bool force_dyn = w.c[RPO_EVOPF_C_FLAGS] != RPO_EVOPF_STATIC_OK;
    const float isx = L.is_extra ? 1.0f : 0.0f;
    float fp[NPV];
    float dsum = 0.0f;                                         // row lanes: sum_p row[NO + p] fp[p] / pivot = (D fp)[mycol]
    int mycol = tid;
    bool solved = false;
    if (!force_dyn) {
        float row[NY];
        jacobian_row<0, NY>(L, v, u, t1, t2, row);
#pragma unroll
        for (int c = 0; c < NY; ++c) row[c] = fmaf(isx, w.vec[c], row[c]);       // lane 28: [g_o^T | g_p^T]
        const float mypiv = gj_static<0, NEQ, 6, 0, NY, TabGrg0>(row);
        solved = pivots_ok<NEQ, 6>(mypiv);
        if (solved) {
            if (L.is_extra) {
#pragma unroll
                for (int p = 0; p < NPV; ++p) w.fp[p] = row[NO + p];             // g_p - D^T g_o (:603-606)
            }
            sync();
#pragma unroll
            for (int p = 0; p < NPV; ++p) fp[p] = w.fp[p];
            float acc = 0.0f;
#pragma unroll
            for (int p = 0; p < NPV; ++p) acc = fmaf(row[NO + p], fp[p], acc);
            dsum = acc / mypiv;
        }
    }
    if (!solved) {                                             // partial pivoting (near-singular Jacobians, or forced)
        float row[NY];
        jacobian_row<0, NY>(L, v, u, t1, t2, row);
#pragma unroll
        for (int c = 0; c < NY; ++c) row[c] = fmaf(isx, w.vec[c], row[c]);
        gj_static<0, 6, 6, 0, NY, TabGrg0>(row);               // the six unit pivots (only the extra row has entries there)
        const float mypiv = gj_dynamic<6, NEQ, 0, NY>(row, mycol);
        if (L.is_extra) {
#pragma unroll
            for (int p = 0; p < NPV; ++p) w.fp[p] = row[NO + p];
        }
        sync();
#pragma unroll
        for (int p = 0; p < NPV; ++p) fp[p] = w.fp[p];
        float acc = 0.0f;
#pragma unroll
        for (int p = 0; p < NPV; ++p) acc = fmaf(row[NO + p], fp[p], acc);
        dsum = acc / mypiv;
    }
    // full gradient by component (:608-610): partial components = fp (lanes 32..46), other components = -(D fp) (row lanes:
    // the unknown of the pivot they own)
    int var = L.own_var;
    if (!solved) {
#pragma unroll
        for (int r = 0; r < NO; ++r) var = (L.is_row && mycol == r) ? kOtherVars[r] : var;
    }
    float gradv = -dsum;
#pragma unroll
    for (int p = 0; p < NPV; ++p) gradv = tid == 32 + p ? fp[p] : gradv;
    if (var < NY) {
        if (dir_out) dir_out[var] = gradv;
        if (apply) {
            const float st = lr * gradv + momentum * w.old[var];
            w.a[var] -= st;
            w.old[var] = st;
        }
    }
    sync();
    return true;
}

// grad_steps (rpo_ddpg.py:266-305, corr_mode 0) on w.a with the lane's own stop test; returns the iteration count
__device__ __forceinline__ int grad_steps_v2(Ws& w, const RowLane& L, int max_steps, float lr, float corr_eps, float momentum) {
    const int tid = lane_id();
    if (tid < NY) w.old[tid] = 0.0f;
    sync();
    int k = 0;
    for (; k < max_steps; ++k)
        if (!grg_iteration_v2(w, L, k == 0, corr_eps, lr, momentum, true, nullptr)) break;
    return k;
}

// Episode data of the loaders (data/demand.py:35-65, data/price.py:29-55) re-keyed on Philox(seed, env, episode):
// hour-t observation (pd[14], qd[14]) into out[0..28) and the price window [t, t+24) into out[33..57) -- all zero when
// t >= 24 (the loaders' `done`, demand.py:71 / price.py:51).
__device__ __forceinline__ void episode_obs(Ws& w, float* out, uint64_t seed, uint32_t env_id, uint32_t episode, int t) {
    RPO_FP_STRICT
    const int tid = lane_id();
    const bool over = t >= T;
    // demand: 11 Exp(1) draws (Dirichlet) + 14 power factors from 7 Philox blocks; word index = tid
    if (tid < 28) {
        const rpo_u4 r = rpo_philox(seed, env_id, episode, RPO_STREAM_EVOPF_DEMAND, (uint32_t)(t * 8 + (tid >> 2)));
        const uint32_t word = (tid & 3) == 0 ? r.x : ((tid & 3) == 1 ? r.y : ((tid & 3) == 2 ? r.z : r.w));
        w.vec[tid] = tid < 11 ? -logf((float)((word >> 8) + 1u) * 5.9604644775390625e-08f)
                              : rpo_u01(word) * (kMaxPf - kMinPf) + kMinPf;
    }
    sync();
    if (tid < NB) {
        float esum = 0.0f;
        for (int j = 0; j < 11; ++j) esum += w.vec[j];
        const int slot = kLoadSlot[tid];
        float ratio = w.c[RPO_EVOPF_C_SHARE + tid];
        if (slot >= 0) ratio += kRho * (w.vec[slot] / esum);
        const float pd = over ? 0.0f : ratio * w.c[RPO_EVOPF_C_PS + (t < T ? t : T - 1)];
        const float pf = w.vec[11 + tid];
        out[tid] = pd;
        out[NB + tid] = over ? 0.0f : pd * tanf(acosf(pf)) * w.c[RPO_EVOPF_C_QSIGN + tid];
    }
    // price: hour h = t + tid of the day (zero past midnight), day = mag * curve * (1 + z)
    if (tid < NAHEAD) {
        const int h = t + tid;
        float v = 0.0f;
        if (!over && h < T) {
            const rpo_u4 r = rpo_philox(seed, env_id, episode, RPO_STREAM_EVOPF_PRICE, (uint32_t)(h >> 1));
            const float z = (h & 1) ? rpo_normal(r.z, r.w) : rpo_normal(r.x, r.y);
            const rpo_u4 m = rpo_philox(seed, env_id, episode, RPO_STREAM_EVOPF_PRICE, (uint32_t)(T / 2));
            const float mag = kRegBias * rpo_normal(m.x, m.y) + 1.0f;
            v = mag * w.c[RPO_EVOPF_C_PRICE + h] * (1.0f + z);
        }
        out[2 * NB + NE + tid] = v;
    }
    sync();
}

}  // namespace rpo_evopf_dev
