// Replay ring sampling (rpo/utils/buffer.py:31-34) on MI355X: uniform-with-replacement index draw (Philox) fused
// with the row gather.  Rows are 96 B (CartSafe, one per 128-byte line of the ring: RPO_CART_RING) / 64 B (SpringPendulum) and
// 16-byte aligned; consecutive lanes copy
// consecutive float4 chunks of a row, so every row is fetched with full 16 B/lane requests and the batch is written
// fully coalesced.  HBM-bound: 2 * row bytes per sample (+8 B index).
#include "common.h"

namespace {

__global__ __launch_bounds__(RPO_BLOCK) void replay_gather_kernel(const float4* __restrict__ rows, int ring_chunks, int chunks_per_row,
                                                                  long long total_chunks,
                                                                  const long long* __restrict__ idx,
                                                                  float4* __restrict__ out) {
    for (long long j = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; j < total_chunks;
         j += (long long)gridDim.x * RPO_BLOCK) {
        const long long b = j / chunks_per_row;
        const int c = (int)(j - b * chunks_per_row);
        out[j] = rows[idx[b] * ring_chunks + c];
    }
}

// NT: the batch is written with non-temporal stores -- large batches (the 2^20-row updates, the 1M-lane clinic) are written once
// and read once by the next launch, a quarter of the LLC: streaming them past the L2 took the launch from 40.2 to 38.6 us at
// 1M rows (0.59 -> 0.62 of 8 TB/s); non-temporal LOADS of the sampled rows made it slower (44.4 us).  Update-sized batches
// keep plain stores: their consumer is the next launch's first instruction.
template <bool NT>
__global__ __launch_bounds__(RPO_BLOCK) void replay_sample_gather_kernel(
    const float4* __restrict__ rows, int ring_chunks, int chunks_per_row, long long total_chunks, long long cap_steps, int n_envs,
    float4* __restrict__ out, long long* __restrict__ idx_out, uint64_t seed, uint32_t salt,
    const long long* __restrict__ ctrl) {
    const long long t = ctrl[RPO_CTRL_T];
    const unsigned long long n_valid = (unsigned long long)((t < cap_steps ? t : cap_steps) * (long long)n_envs);
    for (long long j = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; j < total_chunks;
         j += (long long)gridDim.x * RPO_BLOCK) {
        const long long b = j / chunks_per_row;
        const int c = (int)(j - b * chunks_per_row);
        // np.random.randint(0, size, B) of buffer.py:32 -> counter-based draw, unbiased multiply-shift reduction
        const rpo_u4 r = rpo_philox(seed, (uint32_t)b, (uint32_t)t + salt, RPO_STREAM_SAMPLE, (uint32_t)ctrl[RPO_CTRL_UPDATES]);
        const unsigned long long x = ((unsigned long long)r.x << 32) | r.y;
        const long long row = (long long)__umul64hi(x, n_valid);
        if (NT) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f v = *reinterpret_cast<const v4f*>(&rows[row * ring_chunks + c]);
            __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(&out[j]));
        } else {
            out[j] = rows[row * ring_chunks + c];
        }
        if (idx_out && c == 0) idx_out[b] = row;
    }
}

}  // namespace

extern "C" {

int rpo_replay_gather(const float* rows, int ring_floats, int row_floats, int batch, const long long* idx, float* batch_out,
                      void* stream) {
    if (batch <= 0 || row_floats <= 0 || (row_floats & 3) || ring_floats < row_floats || (ring_floats & 3)) return RPO_ERR_ARG;
    if (!rows || !idx || !batch_out) return RPO_ERR_NULL;
    const int cpr = row_floats / 4;
    const long long total = (long long)batch * cpr;
    hipLaunchKernelGGL(replay_gather_kernel, dim3(rpo_grid_for(total)), dim3(RPO_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(rows), ring_floats / 4, cpr, total, idx, reinterpret_cast<float4*>(batch_out));
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_replay_sample_gather(const float* rows, int ring_floats, int row_floats, long long cap_steps, int n_envs, int batch,
                             float* batch_out, long long* idx_out, unsigned long long seed, unsigned sample_salt,
                             const long long* ctrl, void* stream) {
    if (batch <= 0 || row_floats <= 0 || (row_floats & 3) || ring_floats < row_floats || (ring_floats & 3) || cap_steps <= 0 ||
        n_envs <= 0)
        return RPO_ERR_ARG;
    if (!rows || !batch_out || !ctrl) return RPO_ERR_NULL;
    const int cpr = row_floats / 4;
    const long long total = (long long)batch * cpr;
    if (batch >= 65536)
        hipLaunchKernelGGL(replay_sample_gather_kernel<true>, dim3(rpo_grid_for(total)), dim3(RPO_BLOCK), 0,
                           (hipStream_t)stream, reinterpret_cast<const float4*>(rows), ring_floats / 4, cpr, total, cap_steps, n_envs,
                           reinterpret_cast<float4*>(batch_out), idx_out, (uint64_t)seed, (uint32_t)sample_salt, ctrl);
    else
        hipLaunchKernelGGL(replay_sample_gather_kernel<false>, dim3(rpo_grid_for(total)), dim3(RPO_BLOCK), 0,
                           (hipStream_t)stream, reinterpret_cast<const float4*>(rows), ring_floats / 4, cpr, total, cap_steps, n_envs,
                           reinterpret_cast<float4*>(batch_out), idx_out, (uint64_t)seed, (uint32_t)sample_salt, ctrl);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
