/*
 * rpo_hip.h -- C ABI of librpo_hip.so: the MI355X (gfx950) kernels behind the RPO rollout-collection +
 * constrained-policy-update hot path.
 *
 * The reference (wadx2019/rpo) is pure Python and has NO native / FFI boundary: its seam is the
 * duck-typed Python interface between `rpo/algo` trainers and `rpo/env` constraint oracles.  This header is the
 * C boundary this build puts underneath that interface.  Every entry point names the reference code it replaces
 * (paths relative to the reference root); INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - the caller owns all memory (PyTorch tensors in the shipped host code); nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous and capturable in a hipGraph: all
 *     per-iteration varying quantities (vector-step counter, ring position, Adam step) live in device memory;
 *   - return value: 0 = ok, RPO_ERR_* (<0) = rejected arguments, >0 = hipError_t of the failed launch;
 *   - re-entrant per stream; no hidden global state.
 *
 * Layouts (all float32 unless stated)
 *   CartSafe-v0      state row  [6]  = (x, x_dot, xacc, theta, theta_dot, thetaacc)       cartpole.py:206
 *                    action row [2]  ; basic ("partial") action index = consts.partial     cartpole.py:117
 *                    transition row [24] = s[6] a[2] s'[6] r done eq_viol[1] ineq_viol[6] pad   (ReplayBuffer keys,
 *                                                                       rpo/algo/agent/ddpg_pa.py:70-71)
 *   SpringPendulum-v0 internal row [4] = (theta, theta_dot, l, l_dot)                      pendulum.py:126
 *                    obs row [5] = (cos theta, sin theta, theta_dot, l, l_dot)             pendulum.py:138-140
 *                    transition row [16] = obs[5] a[2] obs'[5] r done eq_viol[1] ineq_viol[1]
 *   replay ring      rows [cap_steps * n_envs, W]; vector step t writes rows (t % cap_steps) * n_envs + lane.
 *   ctrl             int64[RPO_CTRL_LEN]; ctrl[RPO_CTRL_T] = number of vector steps taken so far.  Written only by
 *                    the *_step kernels (last finishing workgroup), read by every kernel that needs t.
 *   stats            float[stats_cap, RPO_STATS_SUB, RPO_STATS_LEN] per-vector-step accumulators, row t % stats_cap;
 *                    the value of a statistic is the sum (max for the *_MAX slots) over the RPO_STATS_SUB sub-rows.
 */
#ifndef RPO_HIP_H
#define RPO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* 3: rpo_split_update gained proj_ws / proj_store_mode / debug (the struct grew); rpo_split_critic_pfront*,
 * rpo_pendulum_project_batchref_ws; rpo_evopf_complete_bwd takes grad_action_b / grad_action2, rpo_evopf_lagrangian takes overwrite;
 * rpo_min_q_bwd, rpo_hw_probe
 * 6: ctrl[RPO_CTRL_NONFINITE] (the step / rollout kernels' sticky failure word) */
#define RPO_ABI_VERSION 6

#define RPO_ERR_ARG (-1)
#define RPO_ERR_NULL (-2)

#define RPO_CART_STATE_DIM 6
#define RPO_CART_ACTION_DIM 2
#define RPO_CART_INEQ 6
#define RPO_CART_ROW 24
#define RPO_CART_RING 32 /* floats between consecutive rows of the REPLAY RING: a 96-byte transition per 128-byte line, so a random
                            row is one line (96-byte spacing: 1.5 lines on average -- gather traffic 1.58 x, round 3); the last
                            8 floats of a ring row are PADDING: the step kernel's full-tile path writes zeros there (whole
                            128-byte lines leave the CU), its partial-tile path leaves them alone, and no kernel reads them.
                            Gathered batches stay RPO_CART_ROW wide.  The stride is compiled into the step / rollout / rider /
                            fused-sampling kernels: a ring must be [rows, RPO_CART_RING] floats (rpo_amd/ops.py refuses others). */
#define RPO_CART_CONSTS_LEN 35 /* C[2] C_p C_o_inv b G[12] d[6] G_r[6] d_r[6], cartpole.py:124-136,397-400 */

#define RPO_PEND_INTERNAL_DIM 4
#define RPO_PEND_OBS_DIM 5
#define RPO_PEND_ACTION_DIM 2
#define RPO_PEND_ROW 16
#define RPO_PEND_RING 16 /* 64-byte rows: two per line, never straddling */

/* EVOPF-v0 (rpo/env/electrical_grid/evopf.py:333-337): observation = pd[14] qd[14] soc[5] price[24]; action = pg[5] qg[5]
 * vm[14] va[14] pe[5]; 14 basic actions = pg at the 4 PV generators, vm at the 5 generator buses, pe[5] (:292-294). */
#define RPO_EVOPF_STATE 57
#define RPO_EVOPF_ACTION 43
#define RPO_EVOPF_PARTIAL 14
#define RPO_EVOPF_EQ 28
#define RPO_EVOPF_INEQ 58
#define RPO_EVOPF_ROW 248 /* state 57 | action 43 | next_state 57 | reward | done | eq_viol 28 | ineq_viol 58 | pad 3 */
/* Constants buffer (device, float32) the host fills from its case tables (rpo_amd/env/electrical_grid/case14.py):
 * Ybus real / imaginary parts row-major (evopf.py:263-266), generator and voltage limits per unit (:251-256), Newton
 * start values (:299-300), cost coefficients already divided by mean(genbase)^2 and multiplied by genbase^(2|1)
 * (obj_fn :509-518), the regularised load curve per hour in per unit of baseMVA (demand.py:48-49), (1 - rho) * nominal
 * load shares and sign(Qd) per bus (demand.py:54-62), the regularised price curve / genbase (price.py:32, evopf.py:57). */
#define RPO_EVOPF_C_YR 0
#define RPO_EVOPF_C_YI 196
#define RPO_EVOPF_C_PMAX 392
#define RPO_EVOPF_C_PMIN 397
#define RPO_EVOPF_C_QMAX 402
#define RPO_EVOPF_C_QMIN 407
#define RPO_EVOPF_C_VMAX 412
#define RPO_EVOPF_C_VMIN 426
#define RPO_EVOPF_C_VM_INIT 440
#define RPO_EVOPF_C_VA_INIT 454
#define RPO_EVOPF_C_QUAD 468
#define RPO_EVOPF_C_LIN 473
#define RPO_EVOPF_C_CONST 478
#define RPO_EVOPF_C_FLAGS 479 /* 1.0f (RPO_EVOPF_STATIC_OK): eliminate in the compiled-in static order of the case14 network (sparse;
                                 falls back to partial pivoting per solve when a pivot is < 2^-6 of the largest one) -- set ONLY by a
                                 host that has checked that every non-zero Ybus entry lies inside case14's branch pattern
                                 (rpo_amd/ops.py EvopfKernels; a skipped non-zero column gives a wrong answer, not a small pivot).
                                 Anything else -- in particular the 0 a table built by other code carries there -- = always
                                 partial pivoting: the safe choice is the default (ABI 5; the meaning was inverted in ABI 4). */
#define RPO_EVOPF_STATIC_OK 1.0f
#define RPO_EVOPF_C_PS 480
#define RPO_EVOPF_C_SHARE 504
#define RPO_EVOPF_C_QSIGN 518
#define RPO_EVOPF_C_PRICE 532
#define RPO_EVOPF_CONSTS_LEN 560

#define RPO_CTRL_LEN 288
#define RPO_CTRL_T 0        /* vector steps completed */
#define RPO_CTRL_ARRIVE 1   /* top-level arrival counter of the running *_step launch (always 0 between launches) */
#define RPO_CTRL_UPDATES 2  /* updates done within the current vector step (host-maintained, 0 unless several updates
                               per step are requested): 4th Philox counter word of the sampling / update-noise draws */
#define RPO_CTRL_NONFINITE 3 /* failure detection (the reference asserts on a NaN action: cartpole.py:170-174,
                               pendulum.py:85-89): 0, or 1 + the vector step at which a lane FIRST stepped with a NaN action or
                               reached a non-finite next state / reward.  Sticky: set once by the *_step / *_rollout / riding
                               kernels (one compare-and-swap per offending lane, nothing on the clean path), cleared only by the
                               host; the transitions of that step ARE in the ring -- the host must stop, not train on. */
#define RPO_CTRL_SUB0 16    /* 16 sub-counters, one per 128-byte line: ctrl[RPO_CTRL_SUB0 + RPO_CTRL_SUB_STRIDE * j] */
#define RPO_CTRL_SUB_STRIDE 16

#define RPO_GRADMAX_SLOTS 16   /* `gradmax` buffers (inf-norm of a gradient slice, clip_grad_norm_) are float[RPO_GRADMAX_LEN]: */
#define RPO_GRADMAX_LEN 512    /* 16 slots 128 bytes apart (slot j at [32 j]), a producing workgroup maxes into slot
                                  (block index) % 16 and the norm is the maximum over the slots -- several hundred atomics on
                                  ONE cache line are served one after the other (2.9 us behind a 448-workgroup backward);
                                  rpo_absmax_slots spreads likewise (rpo_absmax writes word 0); rpo_adam_step* read all slots and zero them (reset_gradmax) */
#define RPO_ADAM_STATE_LEN 544 /* int32 words of an optimiser state buffer (`step_dev` of rpo_adam_step*) */
#define RPO_STATS_SUB 16    /* sub-rows per statistics row: workgroup b adds into sub-row b % 16; the reader sums them */
#define RPO_STATS_LEN 16
#define RPO_STAT_REWARD_SUM 0    /* sum over envs of this step's reward                   rpo_ddpg.py:132 */
#define RPO_STAT_EPISODES 1      /* episodes that ended at this step                       rpo_ddpg.py:134 */
#define RPO_STAT_RETURN_SUM 2    /* sum of their returns                                   rpo_ddpg.py:134 */
#define RPO_STAT_LENGTH_SUM 3    /* sum of their lengths                                                   */
#define RPO_STAT_MAX_INEQ_SUM 4  /* sum over envs of max_i ineq_viol_i                     rpo_ddpg.py:121 */
#define RPO_STAT_MAX_EQ_SUM 5    /* sum over envs of max_i |eq_viol_i|                     rpo_ddpg.py:120 */
#define RPO_STAT_VIOL_COUNT 6    /* envs with max(max_ineq, max_eq) > viol_thresh          cartpole.py:326 */
#define RPO_STAT_MAX_INEQ_MAX 7  /* max over envs                                                          */
#define RPO_STAT_MAX_EQ_MAX 8
#define RPO_STAT_PROJ_ITERS 9    /* sum over envs of GRG iterations taken by the rollout projection       */
#define RPO_STAT_TERMINATED 10   /* episodes ended by the env's own termination test (not the TimeLimit)   */

/* noise_mode of the *_act_project entry points (agent/ddpg_pa.py:101-112, rpo_ddpg.py:98-106) */
#define RPO_NOISE_NONE 0      /* deterministic: ap used as is, not clipped          (take_action(deterministic=True)) */
#define RPO_NOISE_EXPLICIT 1  /* ap += eps * noise[i], then clip to the box         (tests: noise injected)           */
#define RPO_NOISE_PHILOX 2    /* ap += eps * N(0,1) from Philox(seed; env, t), clip (rollout)                         */
#define RPO_NOISE_UNIFORM 3   /* ap  = U(box) from Philox, ignores ap_raw           (warm-up, BoxConstraint.sample)   */
#define RPO_NOISE_CLIP_ONLY 4 /* ap clipped to the box, no noise                    (SAC: noise lives in rsample)     */

/* Philox stream tags (counter word 2) */
#define RPO_STREAM_RESET 1
#define RPO_STREAM_ACT 2
#define RPO_STREAM_SAMPLE 3
#define RPO_STREAM_POLICY 4
#define RPO_STREAM_EVOPF_DEMAND 5 /* episode load profile: Dirichlet + power factors (data/demand.py:53-62) */
#define RPO_STREAM_EVOPF_PRICE 6  /* episode price profile: magnitude + hourly noise   (data/price.py:41-43)  */

int rpo_abi_version(void);

/* Kernel-variant switches: process-wide integers the launch code reads on every call (the library reads NO environment
 * variable).  They select between kernels that compute the same values (bit-equal, or to summation round-off where a test
 * says so) and exist for A/B tests and measurements; the defaults are what is shipped and measured.
 * rpo_tuning(key, value): sets `key` to `value` and returns the previous value; value < 0 only queries; RPO_ERR_ARG for an
 * unknown key.  Not thread-safe against concurrent launches (one host thread per device, as everywhere in this ABI). */
#define RPO_TUNE_FWD_STREAM 0       /* 1: large-n forward (n >= 12288, 128 -> 256 scalar heads) through the weights-in-LDS
                                       streaming kernel (mlp_stream.h); 0: the 64-row tile kernel of rounds 3-4 */
#define RPO_TUNE_FWD_STREAM_WAVES 1 /* waves per workgroup of that kernel: 16 (default, <= 128 registers: 528 us at 2^20 rows) or 12
                                       (<= 168 registers: 576 us) */
#define RPO_TUNE_BWD_ONEPASS 2      /* 1: large-batch backward in one pass over the activations; 0: rows pass + split-K weights pass */
#define RPO_TUNE_GEMM_KSPLIT 3      /* 1: K >= 256 layer launches split k over the four waves of a workgroup; 0: one chain */
#define RPO_TUNE_MLP_GEMM 4         /* 1: 256-wide networks layer by layer (mlp_gemm.h); 0: row-tile kernels */
#define RPO_TUNE_ROLLOUT_WIDE 5     /* fused rollout: 0 16-lane tiles, 1 64-lane tiles, 3 the streaming form (weights stationary in LDS,
                                      rollout_stream.hip; 4: the same with 64-lane groups forced, its form from 2^18 lanes), 2 (default) by lane count: 16-lane tiles below 12 288 lanes,
                                      64-lane tiles below RPO_ROLLOUT_STREAM_FROM, streaming from there */
#define RPO_ROLLOUT_STREAM_FROM 65536
#define RPO_TUNE_BWD_STREAM 6       /* 1: large-batch backward as the two streaming launches of mlp_bwd_stream.h (rows kernel with W0
                                       in LDS + weights kernel); 0: the one-pass / two-pass kernels of round 3 */
#define RPO_TUNE_L1_MFMA 7         /* 1: first layer of the 128-wide row-tile forward as three matrix-core steps (S <= 6, A <= 4);
                                       0: the vector form -- the same fmaf chain, the same bits (rpo_mlp_forward / _multi only) */
#define RPO_TUNE_EVOPF_PLACE 8    /* EXPERIMENT (round 6; default 0 = off): 1: rpo_evopf_act_project places a batch
                                      projection (<= 512 rows) on XCDs 0-1 and a rollout projection on XCDs 2-7 (block b runs on XCD b % 8:
                                      the grid is padded and the other XCDs' workgroups leave at once), so that the two launches that
                                      overlap in the training windows never share a SIMD.  Measured: DESIGN.md 4b */
#define RPO_TUNE_COUNT 9
int rpo_tuning(int key, int value);

/* ---------------------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11), the build's counter-based RNG: key = seed, counter = (id, index, stream, sub);
 * sub = 0 except for the replay-sampling and update-noise draws, where it is ctrl[RPO_CTRL_UPDATES].
 * Fills out[n,4] (uint32) with the raw words for id = id_base + i.  Test hook for bit-exact comparison with
 * oracle/philox.py.  The reference uses global numpy/torch generators instead (scripts/cart_exp.py:9-10), whose
 * streams cannot be reproduced on a GPU; parity tests inject the draws explicitly (SURVEY.md §7 hard part iv).
 * ------------------------------------------------------------------------------------------------------------- */
int rpo_philox_fill(int n, unsigned* out, unsigned long long seed, unsigned id_base, unsigned index,
                    unsigned stream_tag, void* stream);

/* out[i] = N(0,1) (Box-Muller, float32) from Philox(seed; id_base + i, t + salt, stream_tag), t = ctrl[RPO_CTRL_T]
 * (ctrl may be NULL: t = 0).  Replaces the global-generator draws of the update step -- torch.randn_like in
 * PDDDPG_PA.take_action (agent/ddpg_pa.py:109) and Normal.rsample in GaussianSharedPolicy (model/policy.py:59) --
 * with a stream that is reproducible, capturable in a hipGraph and independent of the rank layout. */
int rpo_philox_normal(int n, float* out, unsigned long long seed, unsigned id_base, unsigned salt,
                      unsigned stream_tag, const long long* ctrl, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * CartSafe-v0
 * ------------------------------------------------------------------------------------------------------------- */

/* CartSafeEnv.reset() for every lane (cartpole.py:231-239): state = U(-0.05, 0.05)^6 from
 * Philox(seed; env_id_base + lane, ep_count[lane], RPO_STREAM_RESET); zeroes ep_len / ep_ret. ep_count is kept. */
int rpo_cartsafe_reset(int n_envs, float* state, int* ep_len, float* ep_ret, const unsigned* ep_count,
                       unsigned long long seed, unsigned env_id_base, void* stream);

/* One vectorised env step + TimeLimit + violation bookkeeping + replay scatter + auto-reset, fused.
 * Replaces, per lane: CartSafeEnv.step (cartpole.py:166-229) incl. ineq_dist_np/eq_resid_np of the PRE-step state
 * and UN-clipped action (:229, :410-422), gym TimeLimit(200) (classic_control/__init__.py:9),
 * ReplayBuffer.add (rpo/utils/buffer.py:22-29), the per-step statistics of RPODDPG.run (rpo_ddpg.py:120-145) and
 * the env.reset() that follows a done (:142).
 *   state      [n,6]  in: pre-step state; out: post-step state, or a fresh reset state where done (auto_reset != 0)
 *   action     [n,2]  un-clipped action (clipped to +-10 for the dynamics only)
 *   ep_len/ep_ret/ep_count [n] per-lane episode bookkeeping (updated)
 *   rows       replay ring (may be NULL: no scatter); ring row = (t % cap_steps) * n_envs + lane, t = ctrl[RPO_CTRL_T]
 *   stats      [stats_cap, RPO_STATS_SUB, RPO_STATS_LEN] (may be NULL)
 *   ctrl       int64[RPO_CTRL_LEN]; ctrl[RPO_CTRL_T] is incremented when the launch finishes
 */
int rpo_cartsafe_step(int n_envs, float* state, const float* action, int* ep_len, float* ep_ret, unsigned* ep_count,
                      float* rows, long long cap_steps, float* stats, int stats_cap, long long* ctrl,
                      const float* consts_host, int partial, int max_episode_steps, int auto_reset,
                      float viol_thresh, unsigned long long seed, unsigned env_id_base, void* stream);

/* Exploration noise + box clip + equation solver + GRG projection, fused; one lane per sample, per-lane stop test.
 * Replaces PDDDPG_PA.take_action's noise/clip tail (agent/ddpg_pa.py:108-110) or BoxConstraint.sample (warm-up,
 * model/utils.py:53-62), CartSafeEnv.complete_partial (cartpole.py:369-373) and RPODDPG.grad_steps
 * (rpo_ddpg.py:266-305, corr_mode 0) with CartSafeEnv.eq_resid / ineq_dist / ineq_partial_grad (cartpole.py:375-408).
 *   ap_raw   [n]     actor output (basic action) -- ignored for RPO_NOISE_UNIFORM
 *   noise    [n]     only for RPO_NOISE_EXPLICIT
 *   action   [n,2]   out: completed + projected action
 *   iters    [n]     out (may be NULL): GRG iterations taken by the lane
 *   eps_t = max(eps_end, eps_start - eps_decay * t) (PDDDPG_PA.eps_decay, ddpg_pa.py:118-119), t = ctrl[RPO_CTRL_T]
 *   (ctrl may be NULL: t = 0).  stats may be NULL.
 */
int rpo_cartsafe_act_project(int n, const float* ap_raw, const float* noise, float* action, int* iters,
                             int noise_mode, float eps_start, float eps_end, float eps_decay, float box_lo,
                             float box_hi, int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                             const float* consts_host, int partial, unsigned long long seed, unsigned env_id_base,
                             const long long* ctrl, float* stats, int stats_cap, void* stream);

/* Backward of complete_partial for the actor loss (rpo_ddpg.py:310; autograd through cartpole.py:369-373):
 * grad_ap[i] = grad_action[i,p] - grad_action[i,o] * C_p * C_o_inv. */
int rpo_cartsafe_complete_bwd(int n, const float* grad_action, float* grad_ap, const float* consts_host,
                              int partial, void* stream);

/* Constraint residuals of a batch of full actions: eq_resid [n,1] (cartpole.py:375-376), ineq_resid [n,6] (:378-379).
 * Either output may be NULL.  ineq_dist = max(ineq_resid, 0). */
int rpo_cartsafe_resid(int n, const float* action, float* eq_resid, float* ineq_resid, const float* consts_host,
                       int partial, void* stream);

/* ineq_partial_grad (cartpole.py:396-408) -> step [n,2]. */
int rpo_cartsafe_ineq_partial_grad(int n, const float* action, float* step, const float* consts_host, int partial,
                                   void* stream);

/* Lagrangian term of the actor loss, forward + backward fused (rpo_ddpg.py:312,319,322; Dual.forward dual.py:63-65):
 *   loss_out[0] += scale * sum_b sum_i nu_i * max(0, g_i(a_b));   grad_action[b,:] = scale * sum_i nu_i 1[g_i>0] G_i;
 *   grad_nu[i] += scale * sum_b max(0, g_i(a_b)).   loss_out / grad_nu must be zeroed by the caller. */
int rpo_cartsafe_lagrangian(int n, const float* action, const float* nu, float scale, float* loss_out,
                            float* grad_action, float* grad_nu, const float* consts_host, int partial, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * SpringPendulum-v0 (same roles; the equality is state dependent, pendulum.py:264-288)
 * ------------------------------------------------------------------------------------------------------------- */
int rpo_pendulum_reset(int n_envs, float* internal, float* obs, int* ep_len, float* ep_ret, const unsigned* ep_count,
                       unsigned long long seed, unsigned env_id_base, void* stream);

/* pendulum.py:80-128 + TimeLimit + scatter + auto-reset.  internal [n,4] in/out; obs [n,5] out (may be NULL): the
 * next observation, or the reset observation where done.  The pre-step observation stored in the transition row is
 * recomputed from `internal` (it is bit-identical to what a previous launch wrote to `obs`). */
int rpo_pendulum_step(int n_envs, float* internal, float* obs, const float* action, int* ep_len, float* ep_ret,
                      unsigned* ep_count, float* rows, long long cap_steps, float* stats, int stats_cap,
                      long long* ctrl, int max_episode_steps, int auto_reset, float viol_thresh,
                      unsigned long long seed, unsigned env_id_base, void* stream);

/* noise/clip + complete_partial (pendulum.py:256-262) + grad_steps with the ROW-WISE ineq_partial_grad
 * (pendulum.py:331-343 as evaluated for B = 1; the reference's batched call couples samples).
 * obs rows are read with a row stride of obs_stride floats (>= 5): they may be columns of a gathered batch. */
int rpo_pendulum_act_project(int n, const float* obs, int obs_stride, const float* ap_raw, const float* noise,
                             float* action, int* iters, int noise_mode, float eps_start, float eps_end,
                             float eps_decay, float box_lo, float box_hi, int max_steps, float corr_lr,
                             float corr_eps, float corr_momentum, unsigned long long seed, unsigned env_id_base,
                             const long long* ctrl, float* stats, int stats_cap, void* stream);

#define RPO_PROJ_WS_WORDS 5408
#define RPO_PROJ_WS_GAVE_UP 529
/* complete_partial + the reference's LITERAL batched grad_steps on a training batch (n <= 1024): batch-global stop
 * test (rpo_ddpg.py:271-272) and the sample-coupled ineq_partial_grad of pendulum.py:337-339, grad_i = sum_j
 * 1[a_x,i * dgp_j - bgp_i > 0] * dgp_j.  Used for the TD-target projection of critic_loss
 * (rpo_ddpg.py:330) so that the update step matches the reference; iters_out (int32[1], may be NULL) = iterations. */
int rpo_pendulum_project_batchref(int n, const float* obs, int obs_stride, const float* ap, float* action,
                                  int* iters_out, int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                                  void* stream);
/* The same on one workgroup per 16 rows (n <= 256, max_steps <= 30): the batch's dgp values are all-gathered
 * once per GRG iteration through tagged 8-byte granules in `ws` (RPO_PROJ_WS_WORDS 64-bit words, 128-byte aligned, zero before
 * the first launch; see rpo_split_update.proj_ws for store_mode and the gave-up word).  Same bits as the call above. */
int rpo_pendulum_project_batchref_ws(int n, const float* obs, int obs_stride, const float* ap, float* action, int* iters_out,
                                     int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                                     unsigned long long* ws, int store_mode, void* stream);

/* grad_ap[i] = grad_action[i,0] - grad_action[i,1] * sin/cos (autograd through pendulum.py:256-262). */
int rpo_pendulum_complete_bwd(int n, const float* obs, int obs_stride, const float* grad_action, float* grad_ap,
                              void* stream);

/* eq_resid [n] (pendulum.py:298-300), ineq_resid [n] (:302-303); either output may be NULL. */
int rpo_pendulum_resid(int n, const float* obs, int obs_stride, const float* action, float* eq_resid,
                       float* ineq_resid, void* stream);

int rpo_pendulum_ineq_partial_grad(int n, const float* obs, int obs_stride, const float* action, float* step,
                                   void* stream);

/* g(a) = |a|^2 - 32: loss += scale * nu * sum_b max(0,g); grad_action = scale * nu * 1[g>0] * 2a; grad_nu likewise. */
int rpo_pendulum_lagrangian(int n, const float* action, const float* nu, float scale, float* loss_out,
                            float* grad_action, float* grad_nu, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Replay ring (rpo/utils/buffer.py:3-47)
 * ------------------------------------------------------------------------------------------------------------- */

/* ReplayBuffer.sample (buffer.py:31-34): uniform with replacement over the valid rows
 * n_valid = min(t, cap_steps) * n_envs (t = ctrl[RPO_CTRL_T]); index b = mulhi64(Philox(seed; b, t + sample_salt,
 * RPO_STREAM_SAMPLE).xy, n_valid); copies the row into batch[b, :].  idx_out (int64 [B], may be NULL) receives the
 * drawn indices.  ring_floats = floats between consecutive ring rows (RPO_*_RING; >= row_floats), row_floats = floats of a
 * transition = width of the gathered batch. */
int rpo_replay_sample_gather(const float* rows, int ring_floats, int row_floats, long long cap_steps, int n_envs, int batch,
                             float* batch_out, long long* idx_out, unsigned long long seed, unsigned sample_salt,
                             const long long* ctrl, void* stream);

/* Same gather with caller-provided indices (int64 [B]) -- parity tests inject the reference's np.random.randint. */
int rpo_replay_gather(const float* rows, int ring_floats, int row_floats, int batch, const long long* idx, float* batch_out,
                      void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused TD target + Huber loss, forward + backward (RPODDPG.critic_loss rpo_ddpg.py:327-337,
 * RPOSAC.critic_loss rpo_sac.py:342-353; F.smooth_l1_loss beta = 1, mean reduction)
 *   y_b = r_b + gamma * (1 - done_b) * (min(qn1_b, qn2_b) - alpha * logp_b)        (qn2 / logp NULL: DDPG form)
 *   loss_out[0] += (1/n) sum_b huber(q1_b - y_b) [+ huber(q2_b - y_b)];  grad_q*[b] = clamp(q*_b - y_b, -1, 1) / n
 *   reward / done are read with an element stride (they are columns of the gathered batch).
 * ------------------------------------------------------------------------------------------------------------- */
int rpo_td_huber(int n, const float* q1, const float* q2, const float* qn1, const float* qn2, const float* logp,
                 float alpha, const float* reward, int reward_stride, const float* done, int done_stride, float gamma,
                 float* loss_out, float* grad_q1, float* grad_q2, float* target_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Optimiser plumbing of RPODDPG.train (rpo_ddpg.py:178-205) on flat float32 parameter buffers
 * ------------------------------------------------------------------------------------------------------------- */

/* d(-min(q1, q2))/dq scaled (the actor loss of RPOSAC, rpo_sac.py:335): the smaller one takes the gradient, ties are split like
 * torch.min's backward: w = [q1 < q2] + 0.5 [q1 == q2]; dq1 = w * scale, dq2 = (1 - w) * scale. */
int rpo_min_q_bwd(int n, const float* q1, const float* q2, float scale, float* dq1, float* dq2, void* stream);

/* max_out[0] = max(max_out[0], max_i |x_i|) -- the "inf" norm of clip_grad_norm_ (rpo_ddpg.py:180,193).
 * max_out must hold a non-negative float (0 before the first call of an update). */
int rpo_absmax(long long n, const float* x, float* max_out, void* stream);
/* The same into a gradmax buffer [RPO_GRADMAX_LEN]: workgroup b maxes into slot b % 16 (the norm is the maximum over the
 * slots, as rpo_adam_step* read it) -- a wide launch does not queue its atomics on one word. */
int rpo_absmax_slots(long long n, const float* x, float* gradmax, void* stream);

/* clip_grad_norm_(inf) + torch.optim.Adam step (+ optional DualAdam clamp, model/dual.py:37-45) (+ optional Polyak
 * target update, agent/ddpg_pa.py:77-86), one pass over the flat buffer:
 *   coef = min(1, clip_thres / (max over the slots of gradmax [RPO_GRADMAX_LEN] + 1e-6)) when clip_thres > 0 (gradmax from
 *   rpo_absmax or left by a backward launch); g = coef * grad
 *   (written back, as clip_grad_norm_ does); maximize: g = -g; weight decay; Adam with bias correction at step
 *   step_dev[0] + 1 (the counter is advanced by the launch); clamp_min0: p = max(p, 0); target != NULL:
 *   target = (1 - tau) * target + tau * p.   gradmax is reset to 0 by the launch when reset_gradmax != 0.
 *   zero_grad != 0: instead of the clipped value, 0 is written back -- the gradient is consumed, so that the next
 *   backward pass (which accumulates) needs no separate fill launch (optimizer.zero_grad() folded into the step).
 *   step_dev points at int32[RPO_ADAM_STATE_LEN] (128-byte aligned): {step, the step the cached corrections belong to,
 *   8-byte arrival word (0 between launches),
 *   two doubles: the bias corrections 1 - beta1^(step+1), sqrt(1 - beta2^(step+1)) cached by the previous launch (0.0 = not
 *   cached: they are then computed by every thread; zero-initialise the buffer, and zero the cache when betas change)},
 *   then from word 32 on 16 sub-counters 128 bytes apart (0 between launches): the launch finds its last workgroup
 *   through a two-level arrival tree -- 135 returning atomics on ONE word are served one after the other (3.3 us of a
 *   5.8 us launch, 4.7 us with 270 workgroups); 9 per sub-counter + 16 on the top word are not.
 *   prepared = 1: the caller's previous launch did the bookkeeping (rpo_split_update.prep_step: step_dev[0] already holds
 *   THIS step, the cache its corrections) -- the launch is then purely elementwise: it neither counts its workgroups in
 *   nor touches step_dev, gradmax (reset by the next rpo_split_critic_fwd_a, gradmax_reset) or `clock`. */
int rpo_adam_step(long long n, float* param, float* grad, float* exp_avg, float* exp_avg_sq, int* step_dev,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int maximize, float clip_thres,
                  float* gradmax, int reset_gradmax, int zero_grad, int clamp_min0, float* target, float tau,
                  long long* clock, int prepared, void* stream);
/*   clock (may be NULL): a device counter advanced by one when the launch has finished.  The trainers use it as the
 *   UPDATE clock: the update kernels read their step index from it instead of ctrl[RPO_CTRL_T], so that the next
 *   rollout -- which advances ctrl[RPO_CTRL_T] -- may run concurrently with the update (another stream of the same
 *   hipGraph) without the update seeing the counter move. */

/* Up to four independent optimiser slices in ONE launch (gridDim.y = slice): the tail of a policy step is
 * actor Adam + multiplier DualAdam + the Polyak update of the critic target (rpo_ddpg.py:197-205), three launches of a
 * few microseconds each otherwise.  Every field as in rpo_adam_step; additionally
 *   target2 / n2: a second Polyak target for the first n2 elements of the slice (with a shared state embedding the
 *                 actor's step also moves the critic's copy of it: critic_target[shared] follows in the same pass);
 *   polyak_only:  no optimiser step for this slice, only target = (1 - tau) target + tau param (grad etc. unused).
 * The slices must not overlap. */
typedef struct {
    long long n;
    float *param, *grad, *exp_avg, *exp_avg_sq;
    int* step_dev;
    float lr, beta1, beta2, eps, weight_decay;
    int maximize;
    float clip_thres;
    float* gradmax;
    int reset_gradmax, zero_grad, clamp_min0;
    float* target;
    float tau;
    float* target2;
    long long n2;
    int polyak_only;
    int prepared;                /* as in rpo_adam_step (per slice; with every stepped slice prepared the launch counts no
                                    workgroups in and `clock` is ignored) */
} rpo_adam_seg;
int rpo_adam_step_multi(int count, const rpo_adam_seg* segs, long long* clock, void* stream);

/* soft_update alone (agent/ddpg_pa.py:77-86, sac_pa.py:87-91): target = (1 - tau) * target + tau * param. */
int rpo_polyak(long long n, const float* param, float* target, float tau, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Actor / critic MLPs, exact f32 on the matrix cores (v_mfma_f32_16x16x4_f32 == an fmaf chain).
 * One descriptor covers every network of the reference's agents with hidden_layer = 1 (model/embedding.py:21-29,
 * model/policy.py:24-33,48-57, model/value.py:51-59,125-140):
 *     x0 = s Ws^T + bs (+ a Wa^T + ba; `cat`: [s-part | a-part] concatenated, model/value.py:22-31)
 *     h1 = relu(x0) W0^T + b0 ;  out_k = relu(h1) W1_k^T + b1_k   (k < n_out; second head = SAC's log-std head)
 * Weights are device pointers in torch's nn.Linear layout [out][in]; W0 must be 16-byte aligned.
 * Supported sizes: H = 256 and Ein (= E, or 2E when cat) in {128, 256, 512}; S <= 64, A <= 48 (rpo_mlp_supported).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *Ws, *bs, *Wa, *ba, *W0, *b0, *W1, *b1, *W1b, *b1b;
    int S, A, E, H, n_out, cat;
    int head_dim; /* outputs per head: 0 / 1 = scalar heads (out [n, n_out]); 2..16 = W1_k is [head_dim, H], b1_k
                   * [head_dim], out [n, n_out * head_dim] head-major, raw (out_mode must be 0): the 14 basic actions of
                   * EVOPF-v0, whose state-dependent tanh box is applied by rpo_evopf_act_project (ap_is_raw) */
} rpo_mlp;
typedef struct {
    float *Ws, *bs, *Wa, *ba, *W0, *b0, *W1, *b1, *W1b, *b1b;
    /* Large batches (n >= RPO_SPLITK_FROM rows): NULL, or a scratch buffer of splitk_floats floats.  The weights pass of the
     * backward kernels then splits the batch over up to 256 slices -- every slice accumulates its share of every parameter
     * gradient into its own copy of the network's gradient span inside this buffer (one owner per element, fixed order) and a
     * second launch adds the slices in order: the batch reductions use the width of the chip instead of ~50 workgroups
     * (36 ms -> ~2 ms at 2^20 rows), still bitwise reproducible.  Needs >= 2 x (the span of the gradient tensors, in floats). */
    float* splitk_scratch; long long splitk_floats;
} rpo_mlp_grad;
#define RPO_SPLITK_FROM 16384

/* TD target + Huber loss as a prologue of the critic's backward pass (rpo_ddpg.py:331-335, rpo_sac.py:346-353): with
 * td != NULL rpo_mlp_backward ignores `dout` and computes, per row i,
 *   y = reward + gamma (1 - done) (min(qn1, qn2) - alpha logp),  dq_out[i] = clamp(q[i] - y, -1, 1) / n  (its dout),
 * and loss_partial[i / 16] = that row tile's share of the mean smooth-L1 loss (n_out must be 1).  qn2 / logp may be
 * NULL (RPODDPG).  This is what lets the critic-forward pipelines run Q, Q_targ (and the twin critics) in separate
 * workgroups: the values meet at the kernel boundary instead of inside one workgroup. */
typedef struct {
    const float *q, *qn1, *qn2, *logp;
    const float* reward; int reward_stride;
    const float* done; int done_stride;
    float alpha, gamma;
    float* dq_out;
    float* loss_partial;
} rpo_td;

int rpo_mlp_supported(int E, int H, int cat);

/* Forward of n rows.  s [n, >=S] / a [n, >=A] are read with row strides (columns of a gathered batch are fine).
 * out [n, n_out]; x0_save [n, Ein] / h1_save [n, H] (pre-activations, for rpo_mlp_backward) may be NULL.
 * out_mode 1 applies BoxConstraint's tanh map to output 0: scale * tanh(o) + base (model/utils.py:40-51). */
int rpo_mlp_forward(const rpo_mlp* net_host, int n, const float* s, int s_stride, const float* a, int a_stride,
                    float* out, float* x0_save, float* h1_save, int out_mode, float scale, float base, void* stream);

/* rpo_mlp_forward (out_mode 0) of 1..4 same-shaped scalar-head networks, each on its own inputs, in ONE launch
 * (gridDim.y = network): Q_targ(s', a') and Q(s, a) of a critic update, and RPOSAC's twins (model/value.py:125-140), do
 * not depend on each other.  All arguments are arrays of `count` entries; x0_save[k] / h1_save[k] may be NULL. */
int rpo_mlp_forward_multi(int count, const rpo_mlp* const* nets, int n, const float* const* s, const int* s_stride,
                          const float* const* a, const int* a_stride, float* const* out, float* const* x0_save,
                          float* const* h1_save, void* stream);

/* Column-split forward (rpo_amd/csrc/nsplit_dev.h): the hidden layer of 1..4 same-shaped scalar-head networks (E = 128,
 * H = 256, "add" critics / actors, S <= 6, A <= 4: the first layer is three matrix-core steps) on 16 row tiles x 8 column groups of workgroups each.  Leaves the head
 * partials of the column groups in part_k [8, n, 2]; rpo_mlp_split_head (or the prologue of any consumer kernel) adds
 * them in the order of rpo_mlp_forward's wave loop, so the outputs are bitwise those of rpo_mlp_forward.  This is how
 * the batch-256 update kernels use the width of the chip: one network evaluation is 128 workgroups with a 16 KB weight
 * slice and a 32-instruction MFMA chain each, instead of 16 workgroups with 128 KB and 128 instructions per SIMD. */
int rpo_mlp_split_supported(const rpo_mlp* net_host);
int rpo_mlp_forward_split(int count, const rpo_mlp* const* nets, int n, const float* const* s, const int* s_stride,
                          const float* const* a, const int* a_stride, float* const* part, float* const* x0_save,
                          float* const* h1_save, void* stream);
int rpo_mlp_split_head(const rpo_mlp* net_host, int n, const float* part, float* out, int out_mode, float scale,
                       float base, void* stream);

/* ---- Column-split update of RPODDPG / RPOSAC on CartSafe-v0 / SpringPendulum-v0 (rpo_amd/csrc/nsplit.hip) -------------
 * The batch-256 constrained policy update (rpo_ddpg.py:163-205, rpo_sac.py:167-219) as a chain of SHORT launches that
 * each use the width of the chip (>= 128 workgroups), cut where a value needs every hidden column or every sample:
 *
 *   critic update   fwd_a   ReplayBuffer.sample -> {pi_targ / pi hidden slabs on s' | Q1 (, Q2) hidden slabs on (s, a)}
 *                   fwd_b   head of pi -> (rsample, log pi) -> Complete + Proj -> Q_targ (x2) hidden slabs on (s', a')
 *                           (SpringPendulum: head + the batch-coupled projection are one single-workgroup launch,
 *                           rpo_split_pend_head_project, followed by fwd_b on the projected actions)
 *                   bwd_a   TD target + Huber (from the slab partials) -> {dx0 column groups | dW0 tiles | hidden vectors}
 *                   bwd_b   first-layer gradients (batch reduction of dx0), inf-norm of everything written
 *                   then rpo_adam_step.
 *   policy step     pol_a   pi hidden slabs on s (saved)
 *                   pol_b   head -> noise / rsample + clip -> Complete -> Lagrangian row terms -> Q (x2) slabs on (s, a_pi)
 *                   pol_c   d(-Q or -min Q)/dQ -> critic dx0 column groups -> per-group partials of d/d action
 *                   pol_d   d/d action -> Complete -> head backward -> {actor dx0 column groups | dW0 tiles | vectors}
 *                   pol_e   actor first-layer gradients (+ the critics' dx0 under a shared embedding), multiplier terms
 *                   then rpo_adam_step_multi.
 * Every network is E = 128, H = 256, "add" critic / scalar-head actor (rpo_mlp_split_supported).  All values are
 * bitwise those of the row-tile pipelines (rpo_cartsafe_*_critic_forward + rpo_mlp_backward) for the critic update;
 * the policy step sums d/d action over column groups (agrees to 1e-7 relative).  One struct carries the arguments
 * of all stages; a stage reads only the fields it needs.  part_* are [8, batch, 2] float buffers. */
typedef struct {
    const rpo_mlp *actor, *actor_target, *critic1, *critic2, *critic_target1, *critic_target2;
    const rpo_mlp_grad *critic1_grad, *critic2_grad, *actor_grad;
    int env;                     /* 0 CartSafe-v0 (consts_host / partial), 1 SpringPendulum-v0 */
    int twin;                    /* 0 RPODDPG (actor_target, critic1, critic_target1), 1 RPOSAC (actor, twins) */
    int batch;
    /* ReplayBuffer.sample */
    const float* rows; long long cap_steps; int n_envs;
    float* batch_out; long long* idx_out; const long long* idx_in;
    unsigned long long sample_seed; unsigned sample_salt;
    /* policy draws: rsample of a' (critic update) / take_action noise (policy step); eps_in / noise_in replace Philox */
    const float* eps_in; unsigned long long noise_seed; unsigned noise_id_base, noise_salt;
    const long long* ctrl;
    float scale, base, box_lo, box_hi;
    int max_steps; float corr_lr, corr_eps, corr_momentum;
    const float* consts_host; int partial;
    float alpha, gamma;
    float eps_start, eps_end, eps_decay;          /* exploration schedule of take_action in the actor loss (RPODDPG) */
    /* slab partials and saved pre-activations */
    float *part_pi, *part_q1, *part_q2, *part_qn1, *part_qn2;
    float *x0_1, *h1_1, *x0_2, *h1_2, *x0_a, *h1_a;
    float *logp;                 /* [batch] log pi(a'|s') (critic update) / log pi(a|s) (policy step) */
    float *next_actions;         /* [batch, 2] SpringPendulum: projected a' */
    int *proj_iters;             /* NULL or [1] */
    /* backward */
    float *dq1, *dq2;            /* [batch] dLoss/dQ_k */
    float *loss_partial;         /* [2, ceil(batch / 16)] */
    float *dx0_1, *dx0_2, *dx0_a;
    float *gradmax;              /* NULL or [RPO_GRADMAX_LEN] */
    /* policy step */
    const float* nu; float* nu_grad;
    float *ap_det, *noise_out, *raw, *actions, *g_act, *lag_partial, *lag_out, *da_part, *dout;
    int shared_embedding;
    /* fwd_a: NULL, or the clock / statistics of a rollout launched with defer_clock = 1 right before this update:
     * fwd_a advances rollout_ctrl[RPO_CTRL_T] and clears the next statistics row on its behalf */
    long long* rollout_ctrl; float* rollout_stats; int rollout_stats_cap;
    /* bwd_b: bookkeeping for the rpo_adam_step(prepared = 1) launch of the critic that follows it.  prep_step: NULL, or that
     * optimiser's state buffer -- one thread of bwd_b advances step_dev[0] and leaves the bias corrections of the new step
     * (prep_beta1 / prep_beta2 = the optimiser's betas).  clock_out: NULL, or the update clock, advanced by one (set it
     * when the critic step is the iteration's last optimiser launch).  gradmax_reset (fwd_a): NULL, or the gradmax buffer a
     * prepared Adam launch of the PREVIOUS update consumed: fwd_a zeroes its slots before this update's backward fills them. */
    int* prep_step; float prep_beta1, prep_beta2; long long* clock_out; float* gradmax_reset;
    /* pol_e: the same bookkeeping for up to three slices of the rpo_adam_step_multi launch behind the policy step (actor,
     * multipliers, log alpha); clock_out is advanced by pol_e when set for that stage.  gradmax_reset2 (fwd_a): the actor's. */
    int* prep2_step[3]; float prep2_beta1[3], prep2_beta2[3]; float* gradmax_reset2;
    /* bwd_b / pol_e: NULL, or the update's ctrl buffer: ctrl[RPO_CTRL_UPDATES] += 1 when the stage is the last one of the
     * update that reads it (several updates per vector step: the next update draws with the next sub-index) */
    long long* updates_out;
    /* policy step: head partials of pi(s) between pol_a and pol_b, [8, batch, 2]; NULL: part_pi is used.  A buffer of its
     * own lets pol_a run INSIDE fwd_b's launch (rpo_split_critic_fwd_b_pol) while fwd_b still reads part_pi. */
    float* part_pol;
    /* rpo_split_critic_front: [(3 * ceil(batch / 16) + 1) * 32] words (one 128-byte line per arrival word), 128-byte
     * aligned, zero before the first launch (every launch leaves them zero); word [3 * ceil(batch / 16) * 32] is set to 1 if a
     * workgroup ever gave up waiting for its tile's producers (never, on a healthy device). */
    unsigned* tile_sync;
    /* rpo_split_pend_head_project: NULL (one workgroup), or RPO_PROJ_WS_WORDS 64-bit words, 128-byte aligned, zero before the
     * first launch: the batch-coupled projection then runs on one workgroup per row tile; they all-gather the batch's dgp values once
     * per GRG iteration through tagged 8-byte granules in this buffer (batch <= 256, max_steps <= 30; same bits).  Word
     * RPO_PROJ_WS_GAVE_UP is set to 1 if a workgroup ever gave up waiting for another one's granules (never, on a healthy
     * device).  proj_store_mode: 0 = agent-scope granule stores; 1 = plain stores when the workgroups find themselves on
     * one XCD (checked inside every launch; the polling loads are served by that XCD's L2), agent-scope otherwise. */
    unsigned long long* proj_ws; int proj_store_mode;
    /* tests only (0 in production): bit 0 -- in the fused front launches the policy workgroup of row tile 0, column group 0
     * withholds its hand-over (arrival / granules), so that its consumers run into their bounded wait and raise the gave-up
     * word: the failure path must be loud and must leave the device usable. */
    int debug;
} rpo_split_update;


int rpo_split_critic_fwd_a(const rpo_split_update* u, void* stream);
int rpo_split_critic_fwd_b(const rpo_split_update* u, void* stream);
/* CartSafe-v0 only: rpo_split_critic_fwd_a + rpo_split_critic_fwd_b + rpo_split_critic_bwd_a as ONE launch -- the later
 * stages' workgroups are extra planes of the grid that wait (tile_sync) for the workgroups of their own row tile instead of
 * for a launch boundary (the three stages hand over row-tile-local data only; bwd_b needs every row and stays a launch).  Same
 * values bit for bit (rpo_ddpg.py:163-185 / rpo_sac.py:167-196 up to the TD target's inputs).  Requires that the
 * dispatcher places all workgroups of a row tile on one XCD (rpo_xcc_probe). */
int rpo_split_critic_front(const rpo_split_update* u, void* stream);
/* ... + rpo_split_policy_a as one more plane (policy iteration without a shared embedding, like rpo_split_critic_fwd_b_pol:
 * the caller then skips rpo_split_policy_a; requires part_pol) */
int rpo_split_critic_front_pol(const rpo_split_update* u, void* stream);
/* SpringPendulum-v0: rpo_split_critic_fwd_b + rpo_split_critic_bwd_a as ONE launch in the same way (the batch-coupled
 * projection, rpo_split_pend_head_project, needs every row and keeps fwd_a a launch of its own); _pol: + rpo_split_policy_a. */
int rpo_split_critic_mid(const rpo_split_update* u, void* stream);
int rpo_split_critic_mid_pol(const rpo_split_update* u, void* stream);
/* SpringPendulum-v0: rpo_split_critic_fwd_a + rpo_split_pend_head_project + rpo_split_critic_fwd_b + rpo_split_critic_bwd_a as ONE
 * launch (requires proj_ws, batch <= 256, max_steps <= 30).  The projection is one workgroup per row tile (all on one XCD); what
 * crosses XCDs -- the policy's head partials into it, the projected actions and log pi out of it -- travels as tagged 8-byte
 * granules in proj_ws with agent-scope stores and loads (nothing depends on placement); the row-tile-local hand-overs use
 * tile_sync like rpo_split_critic_front.  Same values bit for bit.  _pol: + rpo_split_policy_a as one more plane. */
int rpo_split_critic_pfront(const rpo_split_update* u, void* stream);
int rpo_split_critic_pfront_pol(const rpo_split_update* u, void* stream);
/* out[x + gx * (y + gy * z)] = the XCD (XCC_ID) workgroup (x, y, z) of a (gx, gy, gz) grid of `threads`-thread workgroups ran
 * on.  rpo_split_critic_front hands data from workgroup to workgroup through ONE XCD's L2; its caller checks with this probe
 * (same grid: 8, ceil(batch / 16), 1 + 3 K; 256 threads) that all workgroups of a row tile share an XCD. */
int rpo_xcc_probe(int gx, int gy, int gz, int threads, int* out, void* stream);
/* Diagnostic (tools/hw_probe.py): out[...] = (XCC_ID << 16) | HW_ID[15:0] (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) of
 * the first wave of every workgroup of a grid whose workgroups hold lds_bytes of LDS and stay resident for `spin` sleep rounds:
 * how many workgroups the dispatcher stacks on one CU / SIMD while others are empty. */
int rpo_hw_probe(int gx, int gy, int gz, int threads, int lds_bytes, int spin, int* out, void* stream);
/* fwd_b + pol_a in one launch (policy iterations, no shared state embedding: the policy slabs on the batch states need
 * nothing the critic update produces); the caller then skips rpo_split_policy_a.  Requires part_pol. */
int rpo_split_critic_fwd_b_pol(const rpo_split_update* u, void* stream);
int rpo_split_pend_head_project(const rpo_split_update* u, void* stream);
int rpo_split_critic_bwd_a(const rpo_split_update* u, void* stream);
int rpo_split_critic_bwd_b(const rpo_split_update* u, void* stream);
int rpo_split_policy_a(const rpo_split_update* u, void* stream);
int rpo_split_policy_b(const rpo_split_update* u, void* stream);
int rpo_split_policy_c(const rpo_split_update* u, void* stream);
int rpo_split_policy_d(const rpo_split_update* u, void* stream);
int rpo_split_policy_e(const rpo_split_update* u, void* stream);
/* rpo_split_policy_a + _b + _c + _d as ONE launch, like rpo_split_critic_front (tile_sync; both envs: the actor loss completes the
 * action without projection steps, rpo_ddpg.py:307-312, so nothing couples the rows); _bc: pol_a already ran inside the
 * critic update's launch (rpo_split_critic_front_pol / _fwd_b_pol / _mid_pol).  Same values bit for bit. */
int rpo_split_policy_front(const rpo_split_update* u, void* stream);
int rpo_split_policy_front_bc(const rpo_split_update* u, void* stream);

/* ---- Rollout stages riding on the critic update's launches ---------------------------------------------------------------
 * Vector step t+1 of the loop (rpo_ddpg.py:93-145) reads nothing the critic update of step t writes when the policy and
 * the critics share no parameters and no policy step lies in between; the update stages above read their own clock
 * (`ctrl` of rpo_split_update), so the two can run side by side.  Instead of a second stream (on one MI355X the fork /
 * join of a hipGraph branch costs more than the 16 us rollout it hides) the rollout's work is handed to workgroups the
 * update launches leave idle:
 *   rpo_split_critic_fwd_a_ride, rpo_split_critic_fwd_b_ride
 *       = the stage + the actor's hidden slabs on the observations of lanes [lane_begin, lane_end) (8 column groups x
 *         lanes / 16 workgroups of 128 threads; partials to `part`).  The forward touches neither the ring nor anything
 *         the update writes, so it may ride anywhere; split over two launches it stays in their shadow.
 *   rpo_split_critic_bwd_b_ride
 *       = the stage + head -> exploration noise / rsample -> box clip -> equation solver -> GRG projection -> env step ->
 *         replay scatter -> statistics -> auto-reset of ALL lanes (one thread per lane, 256 lanes per workgroup);
 *         advances r->ctrl[RPO_CTRL_T].  Must follow rpo_split_critic_fwd_a(_ride) of the same update, which gathers
 *         the batch out of the ring this stage writes into, and forward stages covering every lane.
 * Together they compute exactly what rpo_cartsafe_rollout / rpo_pendulum_rollout compute (same bits: the slab forward is
 * bitwise the row-tile forward and the lane functions are the same code).  u->actor is the rollout policy; u must not
 * have a shared state embedding.  Fields as in rpo_*_rollout. */
typedef struct {
    int n_envs;
    int gauss;                   /* 0: tanh box + exploration noise (RPODDPG), 1: squashed-Gaussian sample (RPOSAC) */
    float scale, base;
    float* state;                /* CartSafe-v0: state [n_envs, 6]; SpringPendulum-v0: internal state [n_envs, 4] */
    float* obs;                  /* SpringPendulum-v0: observation out [n_envs, 5] or NULL; CartSafe-v0: NULL */
    float* action;               /* [n_envs, 2] */
    int* ep_len; float* ep_ret; unsigned* ep_count;
    float* rows; long long cap_steps;             /* replay ring (may be NULL) */
    float* stats; int stats_cap;
    int noise_mode;
    long long* ctrl;             /* the ROLLOUT clock (ctrl[RPO_CTRL_T] = vector steps taken) */
    float eps_start, eps_end, eps_decay, box_lo, box_hi;
    int max_steps; float corr_lr, corr_eps, corr_momentum;
    int max_episode_steps, auto_reset;
    float viol_thresh;
    unsigned env_id_base;
    unsigned long long seed;
    float* part;                 /* [8, n_envs, 2] head partials of the actor between the forward stages and the step */
    int lane_begin, lane_end;    /* forward stages: the lanes of this launch (multiples of 16; lane_end may be n_envs) */
    int defer_clock;             /* step stage: as in rpo_*_rollout -- 1: ctrl[RPO_CTRL_T] is advanced by the next fwd_a */
} rpo_rollout_rider;

int rpo_split_critic_fwd_a_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream);
int rpo_split_critic_fwd_b_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream);
int rpo_split_critic_bwd_b_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream);
/* CartSafe-v0: rpo_split_critic_front with the actor forward of lanes [lane_begin, lane_end) riding in the planes behind its
 * own (replaces fwd_a_ride + fwd_b_ride + bwd_a; bwd_b_ride follows with the lanes' step). */
int rpo_split_critic_front_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream);
/* SpringPendulum-v0: rpo_split_critic_mid with riding planes (replaces fwd_b_ride + bwd_a). */
int rpo_split_critic_mid_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream);
/* SpringPendulum-v0: rpo_split_critic_pfront with riding planes (replaces fwd_a_ride + pend_head_project + mid_ride). */
int rpo_split_critic_pfront_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream);

/* Backward of the same rows given dout [n, n_out] (two launches).  Parameter gradients are ACCUMULATED (+=) into
 * grad_host's buffers (the shared state embedding of shared_param=True receives contributions from two networks,
 * agent/ddpg_pa.py:34-36).  dh [n, H] and dx0 [n, Ein] are caller-provided scratch; da [n, A] (may be NULL) receives
 * the gradient w.r.t. the action input (actor loss: -Q(s, a) back to the policy, rpo_ddpg.py:317).
 * param_grads = 0: only dx0 / da; first_layer_state_only = 1: of the parameters only Ws / bs are accumulated.
 * dh / dx0 are SCRATCH of the two-pass kernels: from RPO_SPLITK_FROM rows the streaming kernels (mlp_bwd_stream.h) keep both in
 * registers and leave the buffers untouched -- a caller must not read them back.
 * gradmax (may be NULL; [RPO_GRADMAX_LEN]): its slots receive max(slot, max |gradient element written by this call|) -- when the buffers were
 * zero before the call this is clip_grad_norm_'s inf-norm of the network (rpo_ddpg.py:180), without the rpo_absmax pass. */
int rpo_mlp_backward(const rpo_mlp* net_host, const rpo_mlp_grad* grad_host, int n, const float* s, int s_stride,
                     const float* a, int a_stride, const float* x0, const float* h1, const float* dout, float* dh,
                     float* dx0, float* da, int param_grads, int first_layer_state_only, float* gradmax,
                     const rpo_td* td, void* stream);

/* rpo_mlp_backward of two networks of the same shape on the same inputs in one pair of launches (SAC's twin critics,
 * model/value.py:125-140): gridDim.y = 2 selects the network.  Results are those of two rpo_mlp_backward calls. */
int rpo_mlp_backward_pair(const rpo_mlp* net1_host, const rpo_mlp_grad* grad1_host, const rpo_mlp* net2_host,
                          const rpo_mlp_grad* grad2_host, int n, const float* s, int s_stride, const float* a,
                          int a_stride, const float* x0_1, const float* h1_1, const float* dout_1, float* dh_1,
                          float* dx0_1, float* da_1, const float* x0_2, const float* h1_2, const float* dout_2, float* dh_2,
                          float* dx0_2, float* da_2, int param_grads, int first_layer_state_only, float* gradmax,
                          const rpo_td* td1, const rpo_td* td2, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * EVOPF-v0 (rpo_amd/csrc/evopf.hip): one wavefront per env lane / batch row; the lane's 22x22 Newton system and the
 * 28x28 block of the equality Jacobian are eliminated in registers, one matrix row per thread (evopf_dev.h).  state [n,57], action [n,43], basic actions [n,14];
 * consts_dev = the RPO_EVOPF_C_* buffer in device memory; state_stride = floats between consecutive state rows (57 for a
 * dense matrix, RPO_EVOPF_ROW when the states are columns of gathered replay rows).  The IEEE-14 bus classification is compiled in, like the
 * reference hard-wires case14 (evopf.py:211,336-337).
 * ------------------------------------------------------------------------------------------------------------- */

/* EVOPFEnv.reset (evopf.py:369-380): hour 0 of episode ep_count[i] (NULL: 0) of lane env_id_base + i; the loaders'
 * random draws (demand.py:53-62, price.py:41-43) come from Philox(seed; env, episode, RPO_STREAM_EVOPF_*). */
int rpo_evopf_reset(int n_envs, float* state, int* ep_len, float* ep_ret, const unsigned* ep_count,
                    const float* consts_dev, unsigned long long seed, unsigned env_id_base, void* stream);

/* EVOPFEnv.step (evopf.py:348-366) + Battery.step (:74-102) + run-loop bookkeeping (rpo_ddpg.py:120-145); same contract
 * as rpo_cartsafe_step: transition rows [RPO_EVOPF_ROW] scattered into the ring, statistics, ctrl[T] advanced by the
 * last workgroup, auto-reset.  An episode ends when the loaders run out of data after 24 hours (demand.py:71). */
int rpo_evopf_step(int n_envs, float* state, const float* action, int* ep_len, float* ep_ret, unsigned* ep_count,
                   float* rows, long long cap_steps, float* stats, int stats_cap, long long* ctrl, int max_episode_steps,
                   int auto_reset, float viol_thresh, const float* consts_dev, unsigned long long seed,
                   unsigned env_id_base, void* stream);

/* take_action's noise + clip to the state-dependent box (agent/ddpg_pa.py:101-112; EVOPFEnv.update evopf.py:769-783)
 * -> complete_partial = Newton power flow (PFFunction.forward, evopf.py:789-855; newton_tol 1e-5, newton_max_iters 50)
 * -> grad_steps (rpo_ddpg.py:266-305, corr_mode 0).  Every row carries its own Newton and GRG stop tests (== the
 * reference's B = 1 rollout calls; on a batch the reference tests the batch maximum, which adds iterations only to
 * rows that had already met the test).  iters [n] (may be NULL) = GRG iterations.  Noise draw of basic action j:
 * Philox(seed; env, ctrl[T], RPO_STREAM_ACT, j / 2) normals (x,y) | (z,w); RPO_NOISE_UNIFORM: word j % 4 of block j / 4.
 * ap_is_raw = 1: ap_raw holds the actor's pre-activation outputs and the kernel applies BoxConstraint's map
 * scale(s) * tanh(.) + base(s) with the lane's own box first (model/utils.py:40-51,75-88). */
int rpo_evopf_act_project(int n, const float* state, int state_stride, const float* ap_raw, const float* noise, float* action, int* iters,
                          int noise_mode, int ap_is_raw, float eps_start, float eps_end, float eps_decay, int max_steps,
                          float corr_lr, float corr_eps, float corr_momentum, float newton_tol, int newton_max_iters,
                          const float* consts_dev, unsigned long long seed, unsigned env_id_base, const long long* ctrl,
                          float* stats, int stats_cap, void* stream);

/* Backward of the actor's output map for EVOPF: ap = clip(scale(s) * tanh(raw) + base(s) + eps_t * noise, lo(s), hi(s))
 * (model/policy.py:24-33 with the state-dependent box, then ddpg_pa.py:108-110); dout [n,14] = dap * d ap / d raw.
 * noise NULL: no noise and no clip. */
int rpo_evopf_tanh_box_bwd(int n, const float* state, int state_stride, const float* raw, const float* noise,
                           float eps_start, float eps_end, float eps_decay, const long long* ctrl, const float* dap,
                           float* dout, const float* consts_dev, void* stream);

/* RPOSAC's policy head on EVOPF: squashed Gaussian with the state-dependent box (model/policy.py:53-66 with
 * BoxConstraint.update_box; the clip of agent/sac_pa.py:111).  raw [n,28] = (mean heads [14] | log-std heads [14]) from
 * rpo_mlp_forward (head_dim 14, n_out 2), eps [n,14]; ap_out [n,14], logp_out [n] (sum over the 14 dimensions, may be
 * NULL); the backward writes draw [n,28] given dap [n,14] and dlogp = coefficient of log pi in the loss. */
int rpo_evopf_gauss_head(int n, const float* state, int state_stride, const float* raw, const float* eps, int deterministic,
                         float* ap_out, float* logp_out, const float* consts_dev, void* stream);
int rpo_evopf_gauss_head_bwd(int n, const float* state, int state_stride, const float* raw, const float* eps,
                             const float* dap, float dlogp, float* draw, const float* consts_dev, void* stream);

/* PFFunction.backward (evopf.py:857-910): grad_ap [n,14] = dL/dz given grad_action [n,43] = dL/dy and the completed
 * action; the Jacobians are re-evaluated at that action (the reference keeps those of the last Newton point). */
int rpo_evopf_complete_bwd(int n, const float* action, const float* grad_action, const float* grad_action_b, const float* grad_action2,
                           float* grad_ap, const float* consts_dev, void* stream);
/*   dL/dy = (grad_action [+ grad_action_b]) [+ grad_action2]: grad_action_b (NULL, or the second twin critic's [n,43] term) and
 *   grad_action2 (NULL, or the Lagrangian's) are added inside the kernel, in that order. */

/* eq_resid [n,28] (evopf.py:520-546) and ineq_resid [n,58] (signed, :548-563); either output may be NULL. */
int rpo_evopf_resid(int n, const float* state, int state_stride, const float* action, float* eq_out, float* ineq_out,
                    const float* consts_dev, void* stream);

/* ineq_partial_grad [n,43] (evopf.py:596-612): reduced gradient of the number of violated inequalities. */
int rpo_evopf_ineq_partial_grad(int n, const float* state, int state_stride, const float* action, float* step_out, const float* consts_dev,
                                void* stream);

/* grad_action [n,43] = J_eq(action)^T grad_eq [n,28] with the entries of eq_jac (evopf.py:614-661): the backward of
 * eq_resid in the Lagrangian baselines' actor loss (ddpg_lag.py:257-263).  autograd_sign != 0 gives the battery columns
 * the sign torch autograd derives from the residual's +pe (evopf.py:532) instead of eq_jac's -I (:639-640). */
int rpo_evopf_eq_vjp(int n, const float* action, const float* grad_eq, float* grad_action, int autograd_sign,
                     const float* consts_dev, void* stream);

/* scale * sum_b sum_j nu_j relu(g_j(s_b, a_b)) accumulated into loss_out, d/d action into grad_action [n,43] (written),
 * d/d nu accumulated into grad_nu [58] (rpo_ddpg.py:312-319, dual.py:63-65). */
int rpo_evopf_lagrangian(int n, const float* state, int state_stride, const float* action, const float* nu, float scale, float* loss_out,
                         float* grad_action, float* grad_nu, const float* consts_dev, int overwrite, void* stream);
/* (overwrite != 0: loss_out is written instead of accumulated -- one launch instead of a fill + this one; grad_nu accumulates) */

/* ---------------------------------------------------------------------------------------------------------------
 * Fused pipelines of one RPO iteration (rpo_amd/csrc/fused.hip): the row-local stages chained
 * inside one workgroup of 16 rows, because at 4096 lanes / batch 256 every launch is dominated by dispatch + cold-start
 * latency.  Bitwise identical to the sequences of single-stage launches they replace.
 * ------------------------------------------------------------------------------------------------------------- */

/* Rollout of one vector step (rpo_ddpg.py:93-145, rpo_sac.py:95-150): actor MLP -> policy head -> complete_partial ->
 * grad_steps -> env step + violations + TimeLimit -> replay scatter -> statistics -> auto-reset.
 * gauss = 0 (RPODDPG, agent/ddpg_pa.py:108-110): the actor's tanh-box output + noise_mode's exploration noise + clip;
 *   == rpo_mlp_forward(actor, out_mode 1) + rpo_<env>_act_project(noise_mode) + rpo_<env>_step (same arguments).
 * gauss = 1 (RPOSAC, agent/sac_pa.py:105-115): actor has the two heads (mean, log-std); the rsample draw of lane i is
 *   normal(philox(seed, env_id_base + i, ctrl[T], RPO_STREAM_POLICY, ctrl[UPDATES])), i.e. rpo_philox_normal(salt 0);
 *   == rpo_philox_normal + rpo_mlp_forward(actor) + rpo_gauss_head + rpo_<env>_act_project(RPO_NOISE_NONE) + step.
 * CartSafe-v0: state [n,6] is the observation.  SpringPendulum-v0: internal [n,4], obs [n,5] (may be NULL: the
 * pipeline itself derives observations from the internal state).
 * defer_clock = 1: the launch does NOT advance ctrl[RPO_CTRL_T] / clear the next statistics row (two dependent atomic
 *   round trips behind its last workgroup, ~1.5 us); the caller's NEXT launch on the stream does it instead --
 *   rpo_split_critic_fwd_a with `rollout_ctrl` set (one plain store, no arrival counting: a later launch starts after
 *   every workgroup of this one has finished). */
int rpo_cartsafe_rollout(const rpo_mlp* actor_host, int gauss, float scale, float base, int n_envs, float* state,
                         float* action, int* ep_len, float* ep_ret, unsigned* ep_count, float* rows, long long cap_steps,
                         float* stats, int stats_cap, long long* ctrl, int noise_mode, float eps_start, float eps_end,
                         float eps_decay, float box_lo, float box_hi, int max_steps, float corr_lr, float corr_eps,
                         float corr_momentum, const float* consts_host, int partial, int max_episode_steps,
                         int auto_reset, float viol_thresh, unsigned long long seed, unsigned env_id_base, int defer_clock,
                         void* stream);
int rpo_pendulum_rollout(const rpo_mlp* actor_host, int gauss, float scale, float base, int n_envs, float* internal,
                         float* obs, float* action, int* ep_len, float* ep_ret, unsigned* ep_count, float* rows,
                         long long cap_steps, float* stats, int stats_cap, long long* ctrl, int noise_mode,
                         float eps_start, float eps_end, float eps_decay, float box_lo, float box_hi, int max_steps,
                         float corr_lr, float corr_eps, float corr_momentum, int max_episode_steps, int auto_reset,
                         float viol_thresh, unsigned long long seed, unsigned env_id_base, int defer_clock, void* stream);

/* Forward half of the critic update (rpo_ddpg.py:165-174, 327-337): ReplayBuffer.sample (Philox draw, or idx_in when
 * given) -> batch_out [B,24]; pi_targ(s') -> Complete + Proj -> Q_targ(s', a') = qn_out; Q(s, a) = q_out with the
 * critic's pre-activations saved (x0_save [B,E], h1_save [B,H]) for rpo_mlp_backward.  Two workgroups per 16-row tile
 * (gridDim.y: target chain | critic) that never talk to each other; the TD target / Huber loss is the prologue of
 * rpo_mlp_backward (rpo_td), where their results meet.
 * == rpo_replay_sample_gather + 3 x rpo_mlp_forward + rpo_cartsafe_act_project. */
int rpo_cartsafe_ddpg_critic_forward(const rpo_mlp* actor_target_host, const rpo_mlp* critic_target_host,
                                     const rpo_mlp* critic_host, float scale, float base, const float* rows,
                                     long long cap_steps, int n_envs, int batch, float* batch_out, long long* idx_out,
                                     const long long* idx_in, unsigned long long seed, unsigned sample_salt,
                                     const long long* ctrl, int max_steps, float corr_lr, float corr_eps,
                                     float corr_momentum, float box_lo, float box_hi, const float* consts_host,
                                     int partial, float* q_out, float* qn_out, float* x0_save, float* h1_save,
                                     void* stream);

/* The policy step of RPODDPG (rpo_ddpg.py:186-205, 307-324) as forward + backward pipelines, env = 0 CartSafe-v0 /
 * 1 SpringPendulum-v0 (E = 128; the actor loss completes the action but does not project it):
 *   forward:  pi(s) with pre-activations saved -> ap = clip(ap_det + eps_t * N(0,1)) (noise_in[b], or the Philox draw of
 *             rpo_philox_normal(noise_id_base, noise_salt, RPO_STREAM_POLICY)) -> Complete -> Q(s, a) saved ->
 *             Lagrangian term: g_act [B,2] = d/d action / B, partial_out [ceil(B/16), 8] = per-workgroup sums of
 *             (nu . relu(g), relu(g_0..5), Q); dq_out = -1/B.
 *   backward: critic rows with da, da += g_act, autograd through Complete and the tanh box / clip, actor rows, the
 *             actor's weights pass (with a shared state embedding the critic's dx0 is added to the actor's first, so one
 *             first-layer reduction yields the embedding's gradient); lag_out = (mean Lagrangian term, mean Q),
 *             nu_grad [6] += mean relu(g); gradmax as in rpo_mlp_backward (actor slice).
 * == rpo_mlp_forward x 2, rpo_philox_normal, rpo_cartsafe_act_project, rpo_cartsafe_lagrangian, rpo_mlp_backward x 2,
 *    rpo_cartsafe_complete_bwd, rpo_tanh_box_bwd and three elementwise launches. */
int rpo_ddpg_actor_forward(int env, const rpo_mlp* actor_host, const rpo_mlp* critic_host, float scale, float base,
                                    float box_lo, float box_hi, float eps_start, float eps_end, float eps_decay,
                                    const float* batch, int batch_size, const float* noise_in, unsigned long long seed,
                                    unsigned noise_id_base, unsigned noise_salt, const long long* ctrl, const float* nu,
                                    const float* consts_host, int partial, float* ap_det, float* noise_out,
                                    float* actions, float* q_out, float* dq_out, float* g_act, float* partial_out,
                                    float* actor_x0, float* actor_h1, float* critic_x0, float* critic_h1, void* stream);
int rpo_ddpg_actor_backward(int env, const rpo_mlp* actor_host, const rpo_mlp_grad* actor_grad_host,
                                     const rpo_mlp* critic_host, int shared_embedding, const float* batch,
                                     int batch_size, const float* actions, const float* g_act, const float* ap_det,
                                     const float* noise, const float* dq, float eps_start, float eps_end,
                                     float eps_decay, float box_lo, float box_hi, float scale, float base,
                                     const long long* ctrl, const float* consts_host, int partial,
                                     const float* actor_x0, const float* actor_h1, const float* critic_x0,
                                     const float* critic_h1, float* actor_dh, float* actor_dx0, float* critic_dh,
                                     float* critic_dx0, float* da, float* dout, const float* partial_in, float* lag_out,
                                     float* nu_grad, float* gradmax, void* stream);

/* The policy step of RPOSAC (rpo_sac.py:191-219, 321-339) as forward + backward pipelines, env = 0 CartSafe-v0 /
 * 1 SpringPendulum-v0 (E = 128; the actor loss completes the action but does not project it, so nothing couples rows):
 *   forward:  two independent workgroups per 16-row tile (gridDim.y = critic): pi(s) heads -> rsample (noise_in[b] or the
 *             Philox draw) + box clip, logp [B] -> Complete -> Q_k(s, a) saved, left in dq_k [B] (a Q value at this point);
 *             role 0 also writes raw [B,2], noise_out, logp, actions, the Lagrangian term's g_act and partial_out
 *             [ceil(B/16), 8] columns 0..6 = sums of (nu . relu(g), relu(g_0..5));
 *   backward: prologue per tile: dq_k <- d(-min(Q1, Q2))/dQ_k / B in place (ties split like torch.min's backward) and
 *             partial[tile][7] = sum(alpha log pi - min Q); then both critics' rows (da_k), da = da1 + da2 + g_act, Complete,
 *             Gaussian head with dlogp = alpha / B, actor rows, the actor's weights pass; lag_out[0] = mean Lagrangian
 *             term, nu_grad +=.  (mean(alpha log pi - min Q) = sum of partial[:, 7] / B, summed by whoever reads the loss.) */
int rpo_sac_actor_forward(int env, const rpo_mlp* actor_host, const rpo_mlp* critic1_host, const rpo_mlp* critic2_host,
                          float scale, float base, float box_lo, float box_hi, float alpha, const float* batch,
                          int batch_size, const float* noise_in, unsigned long long seed, unsigned noise_id_base,
                          unsigned noise_salt, const long long* ctrl, const float* nu, const float* consts_host,
                          int partial, float* raw, float* noise_out, float* logp, float* actions, float* dq1, float* dq2,
                          float* g_act, float* partial_out, float* actor_x0, float* actor_h1, float* critic1_x0,
                          float* critic1_h1, float* critic2_x0, float* critic2_h1, void* stream);
int rpo_sac_actor_backward(int env, const rpo_mlp* actor_host, const rpo_mlp_grad* actor_grad_host,
                           const rpo_mlp* critic1_host, const rpo_mlp* critic2_host, int shared_embedding,
                           const float* batch, int batch_size, const float* actions, const float* g_act, const float* raw,
                           const float* noise, const float* logp, float* dq1, float* dq2, float dlogp, float box_lo,
                           float box_hi, float scale, float base, const float* consts_host, int partial,
                           const float* actor_x0, const float* actor_h1, const float* critic1_x0, const float* critic1_h1,
                           const float* critic2_x0, const float* critic2_h1, float* actor_dh, float* actor_dx0,
                           float* critic1_dh, float* critic1_dx0, float* critic2_dh, float* critic2_dx0, float* da1,
                           float* da2, float* dout, float* partial_in, float* lag_out, float* nu_grad,
                           float* gradmax, void* stream);

/* The same for RPOSAC.critic_loss (rpo_sac.py:342-353): sample -> a' ~ pi(s') with the ONLINE actor (mean / log-std
 * heads, rsample, box clip; the N(0,1) draw of row b is normal(philox(noise_seed, noise_id_base + b, ctrl[T] +
 * noise_salt, RPO_STREAM_POLICY, ctrl[UPDATES])), i.e. rpo_philox_normal, or eps_in[b] when given) -> Complete + Proj ->
 * qn_k = Qk_targ(s', a'), logp_out = log pi(a'|s'); q_k = Qk(s, a) with the pre-activations saved.  Four independent
 * workgroups per 16-row tile (gridDim.y: Q1_targ chain | Q2_targ chain | Q1 | Q2); the TD target
 * y = r + gamma (1 - done) (min(qn1, qn2) - alpha logp) and both Huber terms are the prologue of rpo_mlp_backward_pair
 * (rpo_td).
 * == rpo_replay_sample_gather + rpo_philox_normal + 5 x rpo_mlp_forward + rpo_gauss_head + rpo_cartsafe_act_project. */
int rpo_cartsafe_sac_critic_forward(const rpo_mlp* actor_host, const rpo_mlp* critic_target1_host,
                                    const rpo_mlp* critic_target2_host, const rpo_mlp* critic1_host,
                                    const rpo_mlp* critic2_host, float scale, float base, const float* rows,
                                    long long cap_steps, int n_envs, int batch, float* batch_out, long long* idx_out,
                                    const long long* idx_in, const float* eps_in, unsigned long long sample_seed,
                                    unsigned sample_salt, unsigned long long noise_seed, unsigned noise_id_base,
                                    unsigned noise_salt, const long long* ctrl, int max_steps, float corr_lr,
                                    float corr_eps, float corr_momentum, float box_lo, float box_hi,
                                    const float* consts_host, int partial, float* q1_out, float* q2_out,
                                    float* qn1_out, float* qn2_out, float* logp_out, float* x0_save1, float* h1_save1,
                                    float* x0_save2, float* h1_save2, void* stream);

/* SpringPendulum-v0: the reference's batched projection couples the samples of a batch (pendulum.py:337-339), so the
 * RPOSAC critic-forward chain is cut there into two launches around rpo_pendulum_project_batchref:
 *   front: sample -> batch_out [B,16] -> a' ~ pi(s') -> ap_out [B] (clipped basic action), logp_out [B];
 *   back:  four independent workgroups per tile: qn_k = Qk_targ(s', next_actions [B,2]) | q_k = Qk(s, a) of batch_rows
 *          (pre-activations saved); TD / Huber: rpo_td prologue of rpo_mlp_backward_pair. */
int rpo_pendulum_sac_critic_front(const rpo_mlp* actor_host, float scale, float base, float box_lo, float box_hi,
                                  const float* rows, long long cap_steps, int n_envs, int batch, float* batch_out,
                                  long long* idx_out, const long long* idx_in, const float* eps_in,
                                  unsigned long long sample_seed, unsigned sample_salt, unsigned long long noise_seed,
                                  unsigned noise_id_base, unsigned noise_salt, const long long* ctrl, float* ap_out,
                                  float* logp_out, void* stream);
int rpo_pendulum_sac_critic_back(const rpo_mlp* critic_target1_host, const rpo_mlp* critic_target2_host,
                                 const rpo_mlp* critic1_host, const rpo_mlp* critic2_host, int batch, float* batch_rows,
                                 const float* next_actions, float* q1_out, float* q2_out, float* qn1_out, float* qn2_out,
                                 float* x0_save1, float* h1_save1, float* x0_save2, float* h1_save2, void* stream);

/* The RPODDPG form of the cut pipeline for SpringPendulum-v0 (rpo_ddpg.py:327-337):
 *   front: sample -> batch_out [B,16] -> ap_out [B] = pi_targ(s') (tanh box);
 *   back:  two independent workgroups per tile: qn = Q_targ(s', next_actions) | q = Q(s, a) of batch_rows
 *          (pre-activations saved); TD / Huber: rpo_td prologue of rpo_mlp_backward. */
int rpo_pendulum_ddpg_critic_front(const rpo_mlp* actor_target_host, float scale, float base, const float* rows,
                                   long long cap_steps, int n_envs, int batch, float* batch_out, long long* idx_out,
                                   const long long* idx_in, unsigned long long sample_seed, unsigned sample_salt,
                                   const long long* ctrl, float* ap_out, void* stream);
int rpo_pendulum_ddpg_critic_back(const rpo_mlp* critic_target_host, const rpo_mlp* critic_host, int batch,
                                  float* batch_rows, const float* next_actions, float* q_out, float* qn_out,
                                  float* x0_save, float* h1_save, void* stream);

/* Policy heads around the MLP kernels.
 * DDPG (model/policy.py:30-31, agent/ddpg_pa.py:108-110): ap = clip(ap_det + eps_t * noise), ap_det = scale*tanh(o)+base.
 *   dout[i] = dap[i] * 1[lo <= ap_det + eps_t*noise <= hi] * scale * (1 - tanh(o)^2); noise NULL: no noise, no clip. */
int rpo_tanh_box_bwd(int n, const float* dap, const float* ap_det, const float* noise, float eps_start, float eps_end,
                     float eps_decay, const long long* ctrl, float lo, float hi, float scale, float base, float* dout,
                     void* stream);

/* SAC (GaussianSharedPolicy.forward, model/policy.py:53-66; PDSAC_PA.take_action, agent/sac_pa.py:105-115):
 * raw [n,2] = (mean, log-std head) from rpo_mlp_forward, eps [n] the N(0,1) draw of rsample.
 *   log_std = clamp(raw1 - 3, -23, -2); x = mean + eps * exp(log_std); y = tanh(x);
 *   ap = clip(scale*y + base)  (deterministic: clip(scale*tanh(mean)+base));
 *   log_prob = -eps^2/2 - log_std - log(sqrt(2 pi)) - log(scale*(1 - y^2) + 1e-6). */
int rpo_gauss_head(int n, const float* raw, const float* eps, float scale, float base, float lo, float hi,
                   int deterministic, float* ap_out, float* logp_out, void* stream);

/* Backward of the above: draw [n,2] given dap [n] and a uniform d/d(log_prob) = dlogp (alpha / B in rpo_sac.py:331). */
int rpo_gauss_head_bwd(int n, const float* raw, const float* eps, const float* dap, float dlogp, float scale,
                       float base, float lo, float hi, float* draw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RPO_HIP_H */
