/*
 * dsact.h -- C-ABI of libdsact.so: the MI355X-native (gfx950) DSAC-T off-policy update path.
 *
 * This is the drop-in boundary for ONE path of Jingliang-Duan/DSAC-v2 (all file:line below are
 * relative to the reference repository):
 *
 *   dsac_v2.py:150-206   DSAC_V2.__compute_gradient      -> dsact_compute_grads
 *   dsac_v2.py:320-347   DSAC_V2.__update                -> dsact_apply_update
 *   dsac_v2.py:102-105   DSAC_V2.local_update            -> dsact_step
 *   dsac_v2.py:107-138   get_remote_update_info / remote_update (data-parallel seam)
 *                                                        -> dsact_compute_grads / [all-reduce of the
 *                                                           bound gradient arena] / dsact_apply_update
 *   training/replay_buffer.py:20-50   ReplayBuffer.__init__   -> dsact_buffer_create
 *   training/replay_buffer.py:58-83   store / add_batch       -> dsact_buffer_add
 *   training/replay_buffer.py:85-90   sample_batch            -> dsact_gather (index draw stays on the
 *                                                                host: np.random.randint, bit-exact)
 *   training/trainer.py:72-74         per-key .cuda() of a CPU minibatch -> dsact_load_batch
 *   dsac_v2.py:188-204                the 14 numeric tb_info entries     -> dsact_read_stats
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success, a negative DSACT_E_* on
 *     failure; the message of the last failure on a handle is dsact_last_error(h).
 *   - all device work is enqueued on ONE stream per handle (dsact_set_stream; default: a stream the
 *     handle owns). Calls are asynchronous unless stated; one handle is used from one host thread.
 *   - parameter / optimiser memory is OWNED BY THE CALLER (torch-ROCm tensors on the Python side):
 *     dsact_bind_arenas only records device pointers. Flat fp32 arena order:
 *         online  : q1 | q2 | policy | log_alpha        (dsact_online_count floats)
 *         target  : q1_target | q2_target | policy_target
 *         adam_m, adam_v, grads : same order and size as `online`; `grads` has 2 extra floats at the
 *                   tail (the updated mean_std1/2 EMA, so that ONE all-reduce re-synchronises them)
 *     inside a net: [W0 (out x in, row-major, == nn.Linear.weight) | b0 | W1 | b1 | ...], i.e. the
 *     parameter order of torch's state_dict.
 *     CNN nets (conv_type != 0): [conv0.w | conv0.b | ... | MLP part]. conv weights are stored
 *     [Cout][KH][KW][Cin] (torch's [Cout][Cin][KH][KW] permuted: a dense [Cout x K] matrix whose K order
 *     makes image patches contiguous). The MLP part holds the `mean` and `log_std` MLPs side by side:
 *     layer 0 = [mean.0.weight ; log_std.0.weight] stacked by rows (2H0 x in) | [b_mean ; b_ls];
 *     hidden layer l = mean.W_l (H x H) | log_std.W_l (H x H) | [b_mean ; b_ls];
 *     output layer = (n_out x 2H) matrix [[w_mean, 0], [0, w_ls]] (the zero blocks are structural:
 *     never written by the gradient path, never changed by Adam) | [b_mean ; b_ls].
 *   - all tensors fp32; replay indices int64 on the host API (int32 on device).
 */
#ifndef DSACT_H
#define DSACT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSACT_MAX_HIDDEN_LAYERS 6
#define DSACT_ALGO_DSAC_V2 0
#define DSACT_ALGO_DSAC_V1 1
#define DSACT_CONV_NONE 0
#define DSACT_CONV_TYPE_1 1
#define DSACT_CONV_TYPE_2 2

#define DSACT_OK 0
#define DSACT_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define DSACT_E_HIP (-2)       /* a HIP runtime call failed */
#define DSACT_E_STATE (-3)     /* call order violated (arenas not bound, buffer empty, ...) */
#define DSACT_E_NODEVICE (-4)  /* no MI355X visible */

typedef struct dsact_handle dsact_handle;

/* Hyper-parameters consumed by DSAC_V2.__init__ / ApproxContainer.__init__
 * (dsac_v2.py:27-59,81-90; utils/common_utils.py:48-89). */
typedef struct dsact_config {
  int32_t obs_dim;                              /* obsv_dim */
  int32_t act_dim;                              /* action_dim (<= 32) */
  int32_t n_hidden;                             /* len(hidden_sizes), 1..DSACT_MAX_HIDDEN_LAYERS */
  int32_t hidden[DSACT_MAX_HIDDEN_LAYERS];      /* value/policy_hidden_sizes (same for both) */
  int32_t batch;                                /* replay_batch_size (local batch of this rank) */
  int32_t global_batch;                         /* batch summed over data-parallel ranks (>= batch) */
  int32_t auto_alpha;                           /* auto_alpha */
  int32_t delay_update;                         /* delay_update */
  /* Hyper-parameters cross the boundary as the Python doubles the reference holds: torch.optim.Adam computes its
   * step size and bias corrections, and dsac_v2.py:331 its Polyak factor, in double arithmetic before touching an
   * fp32 tensor -- so does this library, for any value (not only short decimals). */
  double gamma, tau, tau_b;                     /* gamma, tau, tau_b (= tau when absent) */
  double lr_q, lr_pi, lr_alpha;                 /* value/policy/alpha_learning_rate */
  double alpha_fixed;                           /* alpha when !auto_alpha */
  double min_log_std, max_log_std;              /* policy_min/max_log_std */
  double adam_beta1, adam_beta2, adam_eps;      /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 */
  /* CNN approximators (value/policy_func_type == "CNN", networks/cnn.py:151-240,383-461):
   * conv_type 0 = MLP nets; 1 = "type_1" (k 8,4,3 / ch 32,64,64 / stride 4,2,1, hidden 512,256);
   * 2 = "type_2" (k 4,3,3,3,3,3 / ch 8..256 / stride 2,2,2,2,1,1, hidden 256,256,256).
   * With conv_type != 0: obs_dim = img_c*img_h*img_w (replay rows hold the (C,H,W) image), `hidden`
   * are the widths of the `mean` / `log_std` MLPs that follow the conv stack. */
  int32_t conv_type;
  int32_t img_c, img_h, img_w;
  /* algo 0 = DSAC_V2 (DSAC-T: twin critics, mean_std refinements); 1 = DSAC_V1 (reference dsac_v1.py: ONE critic
   * `q` + `q_target`, fixed TD_bound, variance-weighted critic pseudo-loss :217-226). With algo 1 the arenas are
   * online = q | policy | log_alpha and target = q_target | policy_target; MLP nets only. */
  int32_t algo;
  double td_bound;                              /* TD_bound (DSAC_V1 only; reference default 20) */
  int32_t v1_unbounded;                         /* DSAC_V1 `bound` kwarg NEGATED (0 = the reference default bound=True, the
                                                 * variance-weighted pseudo-loss of dsac_v1.py:217-226; 1 = bound=False, the plain
                                                 * Gaussian negative log-likelihood of :227-228) */
  /* value_hidden_activation / policy_hidden_activation (utils/common_utils.py:16-45): 0 gelu (every shipped example),
   * 1 relu, 2 elu, 3 selu, 4 sigmoid, 5 tanh -- torch's default arguments. Output activations are linear. */
  int32_t value_act, policy_act;
  /* policy_act_distribution (utils/act_distribution_cls.py): 0 = "TanhGaussDistribution" (:21-79, every shipped example),
   * 1 = "GaussDistribution" (:82-115: no squashing -- action = mean + std * eps, log-prob of the plain diagonal Gaussian;
   * the action limits are only used by the caller's clipping and by mode()). */
  int32_t act_dist;
  /* policy_std_type (networks/mlp.py:43-73): 0 = "mlp_shared" (one MLP -> mean | log_std; every shipped example),
   * 1 = "parameter" (:63-73,92-97: the MLP gives the mean, log_std is a learnable (1, act_dim) parameter). With 1 the arenas
   * keep the (2 act_dim x H) output layer: rows [act_dim, 2 act_dim) of its weight are structurally zero (the caller zeroes
   * them once; their gradient is masked, so Adam / Polyak leave them at 0) and the second half of its bias IS log_std.
   * DSAC_V2 with MLP nets, on BOTH kernel families: the row-slice chains (equal hidden widths 64 / 128 / 256, batch a multiple
   * of 16; DwProb::msplit masks the gradient) and the tile stages (every other shape, incl. split-K at batch > 448, unequal
   * widths and non-linear output activations; GemmProb::mzero) -- tests/test_std_parameter.py covers a shape of each. */
  int32_t policy_std_param;
  /* value_output_activation / policy_output_activation (utils/common_utils.py:16-45 -> networks/mlp.py:15-20: the module that
   * follows the LAST Linear): 0 = "linear" (every shipped example), 1 relu, 2 elu, 3 selu, 4 sigmoid, 5 tanh, 6 gelu (whose
   * derivative is no function of its output: the tile-stage heads store it beside the output, and a handle with it stays on the
   * tile-stage kernels). DSAC_V2 with MLP nets only. Round 6: served by the row-slice chains wherever a linear head would
   * be (the generic-activation instantiations of the forward kernels apply it in the heads, the backward row phases multiply by
   * its derivative expressed through the stored POST-activation outputs) and by both acting forwards; shapes the chains do not
   * take, and batch >= 1024's throughput-regime kernels, fall to the chains' generic forms / the tile-stage kernels as for linear
   * heads. With policy_std_param the policy's activation applies to the mean half only (log_std is a plain parameter,
   * networks/mlp.py:92-97). */
  int32_t value_out_act, policy_out_act;
  /* value_hidden_sizes != policy_hidden_sizes (utils/common_utils.py:59-62 reads them per key): `hidden` sizes the critics,
   * `policy_hidden[l]` > 0 the policy nets (all zeros: the same widths); policy_n_hidden below for another number of layers.
   * DSAC_V2 with MLP nets on the tile-stage kernels (the row-slice chains run one width per layer across every unit -- a caller
   * who wants such nets on the chains stores them zero-padded to a common width and passes THAT as `hidden`: the Python host
   * side does, dsac-v2_amd/dsact/layout.py ArenaLayout pad_to). */
  int32_t policy_hidden[DSACT_MAX_HIDDEN_LAYERS];
  /* policy_std_type "mlp_separated" (networks/mlp.py:46-57,80-85): 1 = the policy is TWO MLPs over the observation, `mean` and
   * `log_std`, each obs -> hidden -> act_dim (0: one of the two forms policy_std_param selects). The arenas keep them side by
   * side, exactly as the CNN nets' twin trunks: layer 0 one dense (2 H0 x obs) matrix [mean.0 ; log_std.0], hidden layer l two
   * (H x Hprev) blocks mean | log_std, output layer the dense (2 act_dim x 2 H) matrix [[w_mean, 0], [0, w_log_std]] whose two
   * zero blocks are structural (the caller zeroes them once; no gradient is ever written there), biases [b_mean ; b_log_std].
   * DSAC_V2 with MLP nets, tile-stage kernels (the critics stay single MLPs: the row-slice chains run one trunk count per
   * launch); policy_std_param must be 0. Both acting forwards serve it (a block layer is a row range with an input offset). */
  int32_t policy_twin;
  /* len(policy_hidden_sizes) when it differs from len(value_hidden_sizes) (0: the same number of layers). The policy nets then
   * take `policy_hidden[0 .. policy_n_hidden)` (every entry > 0) and their own depth everywhere; the critics keep `n_hidden` /
   * `hidden`. DSAC_V2 with MLP nets, tile-stage kernels (stage lists, heads and weight-gradient tiles are built per net). */
  int32_t policy_n_hidden;
} dsact_config;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int dsact_version(void);
int dsact_device_count(void);
int dsact_create(const dsact_config* cfg, int device, dsact_handle** out);
int dsact_destroy(dsact_handle* h);
const char* dsact_last_error(const dsact_handle* h);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL -> handle-owned stream */
int dsact_set_stream(dsact_handle* h, void* hip_stream);
/* the hipStream_t every call on this handle enqueues on -- host code that interleaves its own device work
 * (torch.distributed collectives on the gradient arena, dsac_v2.py:107-138 seam) must issue it on THIS stream
 * (torch.cuda.ExternalStream) or order against it with events; never NULL after dsact_create */
void* dsact_get_stream(const dsact_handle* h);
int dsact_sync(dsact_handle* h); /* blocks until the handle's stream is idle */

/* ---- parameters (ApproxContainer, dsac_v2.py:19-62) ------------------------------------------- */
size_t dsact_online_count(const dsact_handle* h); /* floats in q1|q2|policy|log_alpha */
size_t dsact_target_count(const dsact_handle* h); /* floats in q1_t|q2_t|policy_t */
size_t dsact_q_count(const dsact_handle* h);      /* floats in one Q net */
size_t dsact_pi_count(const dsact_handle* h);     /* floats in the policy net */
int dsact_bind_arenas(dsact_handle* h, float* online, float* target, float* adam_m, float* adam_v,
                      float* grads /* online_count + 2 floats */);
/* act_high_lim / act_low_lim buffers of StochaPolicy (networks/mlp.py:75-76); host pointers */
int dsact_set_action_limits(dsact_handle* h, const float* high, const float* low);
/* Adam step counters (q, policy, alpha) and the mean_std EMA state (dsac_v2.py:88-89,233-241);
 * mean_std < 0 encodes the reference's -1.0 "not yet initialised" sentinel. Synchronous.
 * dsact_set_state (and a successful dsact_bind_arenas) also ACKNOWLEDGES a hand-over timeout: after one (DSACT_E_HIP from
 * any entry point, see dsact_debug_set below) parameters / moments / targets are suspect and every update entry point
 * fails with DSACT_E_STATE until the caller has restored them and called one of the two
 * (dsact_debug_get(h, "state_invalid") reads the flag). */
int dsact_get_state(dsact_handle* h, int32_t adam_steps[3], float mean_std[2]);
int dsact_set_state(dsact_handle* h, const int32_t adam_steps[3], const float mean_std[2]);
/* The reference re-reads its `adjustable_parameters` (dsac_v2.py:92-99: gamma, tau, auto_alpha, alpha, delay_update;
 * dsac_v1.py adds TD_bound) on every update, so assigning one between updates takes effect on the next. Same here:
 * the value is used from the next enqueued update on; a captured graph (dsact_graph_build) has the old value baked
 * in and is dropped -- build it again. tau also sets tau_b when `also_tau_b` semantics apply (DSACT_HYPER_TAU keeps
 * tau_b untouched; set DSACT_HYPER_TAU_B explicitly). */
#define DSACT_HYPER_GAMMA 0
#define DSACT_HYPER_TAU 1
#define DSACT_HYPER_TAU_B 2
#define DSACT_HYPER_AUTO_ALPHA 3
#define DSACT_HYPER_ALPHA 4
#define DSACT_HYPER_DELAY_UPDATE 5
#define DSACT_HYPER_V1_BOUND 7      /* DSAC_V1 `bound` (0 / 1) */
#define DSACT_HYPER_TD_BOUND 6
int dsact_set_hyper(dsact_handle* h, int32_t which, double value);

/* ---- replay buffer (training/replay_buffer.py) -------------------------------------------------- */
int dsact_buffer_create(dsact_handle* h, int64_t capacity);
/* n transitions, SoA host arrays (obs[n*O], act[n*A], rew[n], obs2[n*O], done[n], logp[n]); ring
 * write at ptr with the reference's ptr/size semantics (replay_buffer.py:58-79). ASYNCHRONOUS: the arrays are copied
 * into a pinned staging slot before the call returns (the caller may reuse them at once), one H2D copy and the ring
 * write are enqueued on the handle's stream and nothing waits for them; every later call on the handle is ordered
 * behind them. dsact_buffer_size / dsact_buffer_ptr reflect the add immediately. */
int dsact_buffer_add(dsact_handle* h, int64_t n, const float* obs, const float* act, const float* rew,
                     const float* obs2, const float* done, const float* logp);
int64_t dsact_buffer_size(const dsact_handle* h);
int64_t dsact_buffer_ptr(const dsact_handle* h);
/* bulk fill of rows [row0,row0+n) directly from DEVICE arrays (synthetic benchmark buffers);
 * sets size=max(size,row0+n), ptr=(row0+n)%capacity */
int dsact_buffer_fill_device(dsact_handle* h, int64_t row0, int64_t n, const float* obs, const float* act,
                             const float* rew, const float* obs2, const float* done);
/* gather rows idx[0..batch) into the handle's minibatch staging area (== sample_batch + .cuda()) */
int dsact_gather(dsact_handle* h, const int64_t* idx_host, int32_t batch);
/* copy the staged minibatch back to host arrays (any may be NULL); synchronous. `logp` comes from
 * the ring (it is not part of the staging area the update reads). */
int dsact_read_batch(dsact_handle* h, float* obs, float* act, float* rew, float* obs2, float* done,
                     float* logp);
/* stage a minibatch produced elsewhere (reference ReplayBuffer + new algorithm mix). Each pointer may be a HOST
 * array (the reference's CPU batch) or a DEVICE array on the handle's GPU (the reference trainer's `.cuda()`
 * tensors, training/trainer.py:72-74: no round trip through the host); the copies are issued on the handle's
 * stream and completed before the call returns, so the caller must have finished writing the sources. */
int dsact_load_batch(dsact_handle* h, const float* obs, const float* act, const float* rew,
                     const float* obs2, const float* done);
/* index table for graph replay: rows x batch indices, row r is consumed by the r-th replayed step */
int dsact_upload_index_table(dsact_handle* h, const int64_t* idx_host, int32_t rows);

/* ---- noise (the torch.randn draws of one __compute_gradient) ------------- */
/* parity mode: inject eps_new[B*A], eps_2[B*A], z5[B], z6[B] (host pointers) for the next step */
int dsact_set_noise(dsact_handle* h, const float* eps_new, const float* eps_2, const float* z5,
                    const float* z6);
/* production mode: device Philox4x32-10 + Box-Muller keyed by (seed, iteration); 0 disables */
int dsact_set_device_rng(dsact_handle* h, uint64_t seed);

/* ---- the update ----------------------------------------------------------------------------------- */
#define DSACT_F_DATA_PARALLEL 2u           /* dsact_graph_build: capture gather -> grads -> all-reduce (dsact_comm_init) -> apply */
#define DSACT_F_SKIP_ACTOR_ON_OFF_ITERS 1u /* "fast": skip actor/alpha backward when it % delay != 0
                                              (their gradients are discarded by the reference,
                                              dsac_v2.py:174-186 vs :324); parameter trajectory is
                                              unchanged */
int dsact_compute_grads(dsact_handle* h, int64_t iteration, uint32_t flags);
int dsact_apply_update(dsact_handle* h, int64_t iteration);
int dsact_step(dsact_handle* h, int64_t iteration, uint32_t flags); /* compute_grads + apply_update */
/* hipGraph path: captures `steps_per_graph` consecutive updates (gather from the index table +
 * step, iteration read from device state) and replays them; dsact_graph_run enqueues n_steps
 * updates starting at `first_iteration` (n_steps % steps_per_graph == 0). The replay ring must not
 * be written while a graph runs: inside a graph the minibatch of update s+1 is gathered while
 * update s is still in flight (it rides in that update's loss launch); results are bit-identical
 * to the same updates issued one by one. The last update's minibatch stays staged.
 * Pipelined graph (row-slice chains, batch <= 256, device RNG or dsact_run_group's noise table, 2 <= delay_update <= 4,
 * steps_per_graph >= 2; DSACT_F_SKIP_ACTOR_ON_OFF_ITERS: the same graph without the discarded policy backward):
 * policy, log_alpha and the three target nets change only when iteration % delay_update == 0 (dsac_v2.py:320-347), so
 * update it + 1 of such a window sees the policy update `it` saw -- the forward launch of update `it` also evaluates
 * policy(obs) + rsample and policy_target(obs2) for the NEXT minibatch (gathered two updates ahead) and update it + 1's
 * forward launch holds only the chains that need the fresh critics. One graph per phase first_iteration % delay_update is
 * captured; dsact_graph_run picks per replay. Same bits as eager updates. DSACT_NO_PIPE=1 captures the plain graph;
 * dsact_debug_get(h, "pipe_graph") tells which one was captured. */
int dsact_graph_build(dsact_handle* h, int32_t steps_per_graph, uint32_t flags);
int dsact_graph_run(dsact_handle* h, int64_t first_iteration, int64_t n_steps);
/* OffSerialTrainer.step between two sampler calls (training/trainer.py:63-82 with sample_interval = n_steps; the reference's
 * CNN examples run 8, example_train/dsacv2_cnn_carracing_offasync.py:133): n_steps x { ReplayBuffer.sample_batch
 * (replay_buffer.py:85-90) -> DSAC_V2.local_update (dsac_v2.py:102-105) } as ONE graph replay. While no add_batch intervenes
 * the ring size is constant and only np.random.randint consumes the NumPy stream (replay_buffer.py:86), so the caller
 * draws the n_steps index rows up front with the reference's own calls: idx[n_steps][batch]. noise (nullable): the
 * reference's torch.randn draws of those updates, [n_steps][2*batch*act_dim + 2*batch] floats =
 * eps_new | eps_2 | z5 | z6 per update -- strict RNG through the (pipelined) graph; NULL: device Philox keyed by the
 * iteration (dsact_set_device_rng). Asynchronous: pinned staging, a stream-ordered reset of the replay counters, no host
 * wait. Captured graphs are kept per (n_steps, flags, noise mode); the first group of a new shape pays its capture.
 * Same bits as n_steps x { dsact_gather; dsact_step }. The ring must not be written while the group runs (dsact_sync or
 * any synchronising call first) -- dsact_buffer_add is stream-ordered behind it and therefore safe. */
int dsact_run_group(dsact_handle* h, int64_t first_iteration, int32_t n_steps, const int64_t* idx, const float* noise, uint32_t flags);

/* data-parallel replay (one process per GPU): the same two halves with iteration / index-table row
 * taken from device state, so the host never touches the step. Between them the caller all-reduces
 * the bound gradient arena (online_count + 2 floats, average over ranks) on the same stream:
 *   dsact_dp_enqueue_grads  = gather(table row) + __compute_gradient   (get_remote_update_info)
 *   dsact_dp_enqueue_apply  = __update                                   (remote_update)
 * dsact_dp_begin sets the first iteration (synchronous). */
int dsact_dp_begin(dsact_handle* h, int64_t first_iteration);
int dsact_dp_enqueue_grads(dsact_handle* h, uint32_t flags);
/* dsact_dp_enqueue_grads in two halves so that the all-reduce of the critics' segment overlaps the actor's
 * backward: after _critic the q1|q2 segment of the gradient arena is final; _actor fills policy | log_alpha |
 * mean_std tail. [all_reduce(q1|q2) async] ... [all_reduce(rest)] ... dsact_dp_enqueue_apply. */
int dsact_dp_enqueue_grads_critic(dsact_handle* h, uint32_t flags);
int dsact_dp_enqueue_grads_actor(dsact_handle* h, uint32_t flags);
/* STRICT data-parallel mode: the mean_std EMA of every rank uses the GLOBAL batch
 * mean of the critics' std, which needs one 2-float all-reduce between the forward and the loss:
 *   dsact_dp_set_strict(h, buf)      buf = 2 device floats owned by the caller (a torch tensor the caller
 *                                    all-reduces with SUM); NULL switches back to the one-collective mode
 *   dsact_dp_enqueue_forward         gather + forward + local {sum std1, sum std2} -> buf
 *   [all_reduce(buf, SUM)]
 *   dsact_dp_enqueue_backward        loss (mean over cfg.global_batch) + backward -> gradient arena
 *   [all_reduce(grads) / world]      then dsact_dp_enqueue_apply as in the one-collective mode */
int dsact_dp_set_strict(dsact_handle* h, float* std_sums_dev);
int dsact_dp_enqueue_forward(dsact_handle* h, uint32_t flags);
int dsact_dp_enqueue_backward(dsact_handle* h, uint32_t flags);
int dsact_dp_enqueue_apply(dsact_handle* h);

/* ---- native collective (RCCL over xGMI) ------------------------------------------------------------------
 * The all-reduce of the data-parallel seam (dsac_v2.py:107-138: gradients out of get_remote_update_info, into
 * remote_update) issued by the library itself on the handle's stream, so that a whole data-parallel update
 * -- gather -> gradients -> all-reduce -> Adam/Polyak -- is ONE hipGraph (BASELINE.json configs[4]) with no
 * per-step host call. librccl is opened at run time (`rccl_path`: the copy torch already loaded, so that one
 * RCCL serves the process; NULL: "librccl.so").
 *   dsact_comm_unique_id   rank 0 creates the 128-byte ncclUniqueId; the caller ships it to the other ranks
 *   dsact_comm_init        every rank joins (ncclCommInitRank); the communicator lives in the handle
 *   dsact_dp_enqueue_allreduce   average of the gradient arena (+ the 2-float mean_std tail) over the ranks,
 *                          enqueued between dsact_dp_enqueue_grads and dsact_dp_enqueue_apply
 *   dsact_graph_build(..., DSACT_F_DATA_PARALLEL)  captures that chain per step; in strict mode
 *                          (dsact_dp_set_strict) also the 2-float all-reduce of the std sums before the loss */
int dsact_comm_unique_id(const char* rccl_path, uint8_t id[128]);
int dsact_comm_init(dsact_handle* h, int32_t rank, int32_t world, const uint8_t id[128], const char* rccl_path);
int dsact_comm_destroy(dsact_handle* h);
int dsact_dp_enqueue_allreduce(dsact_handle* h);

/* 14 numeric tb_info entries of the last update in the order of dsac_v2.py:188-202
 * (avg_q1, avg_q2, avg_std1, avg_std2, min_std1, min_std2, loss_actor, loss_critic, policy_mean,
 *  policy_std, entropy, alpha, mean_std1, mean_std2) + out[14] = iteration, out[15] = device milliseconds of the
 *  last dsact_step / dsact_compute_grads (stream events around it; -1 when the last update was a graph replay or a
 *  snapshot slot is read). With a gradient computed and not yet applied (dsact_compute_grads without
 *  dsact_apply_update) mean_std1/2 are the values that gradient's loss used, as the reference reports them.
 * Synchronous (this is the only per-step host sync, and only when the trainer logs). */
int dsact_read_stats(dsact_handle* h, float out[16]);
/* A reference-style caller may keep the tb_info dict of update k and read it after update k+1 has been issued
 * (deferred logging). dsact_stats_snapshot reduces the LAST update's statistics into ring slot `slot`
 * (0 .. DSACT_STATS_SLOTS-1) asynchronously -- one 64-thread launch, no host sync; dsact_stats_read copies a slot
 * out (synchronous). Same 16 floats as dsact_read_stats. */
#define DSACT_STATS_SLOTS 16
int dsact_stats_snapshot(dsact_handle* h, int32_t slot);
int dsact_stats_read(dsact_handle* h, int32_t slot, float out[16]);

/* ---- measurement / debugging -------------------------------------------------------------------- */
/* time n replays of the step on the handle's stream with hipEvents: total milliseconds */
int dsact_time_steps(dsact_handle* h, int64_t first_iteration, int64_t n_steps, uint32_t flags,
                     int32_t use_graph, float* ms_total);
/* hipEvent timing of `reps` back-to-back launches of ONE forward tile stage (k_stage<KC,KC,GELU>,
 * stage index 0..2*n_hidden-1: group A layers then group B layers) on the handle's stream; also
 * returns the stage's algorithmic multiply-accumulate count. Leaves the activations of that stage
 * overwritten (measurement only). */
int dsact_time_stage(dsact_handle* h, int32_t stage, int32_t reps, float* ms_total, double* macs);
/* 1 when this handle runs the update's MLP layers as row-slice fused chains (csrc/dsact_chain.h: DSAC_V2 / DSAC_V1 with equal
 * hidden widths of 64 / 128 / 256 and batch % 16 == 0 -- MLP nets, or the twin mean / log_std trunks of the CNN nets behind
 * their conv stacks at batch <= 1024; DSACT_NO_CHAIN=1 / DSACT_NO_CHAIN_CNN=1 force the per-layer tile stages), else 0.
 * Both paths replace the same reference functions (dsac_v2.py:150-347) and fill the same debug buffers. */
int dsact_chain_active(const dsact_handle* h);
/* per-kernel hipEvent timing of ONE eager step: fills up to `cap` entries; returns count in *n */
typedef struct dsact_kernel_time {
  char name[32];
  float ms;
  int32_t blocks;
} dsact_kernel_time;
int dsact_profile_step(dsact_handle* h, int64_t iteration, uint32_t flags, dsact_kernel_time* out,
                       int32_t cap, int32_t* n);
/* the same for n_steps consecutive updates issued eagerly as the launch sequence dsact_graph_build(n_steps, flags) would
 * capture (the pipelined sequence when that is what it would capture: its forward launches are named "chain_fwd",
 * "chain_fwd+next", "chain_fwd_q", "chain_fwd_q+next" by what they hold) */
int dsact_profile_steps(dsact_handle* h, int64_t first_iteration, int32_t n_steps, uint32_t flags, dsact_kernel_time* out,
                        int32_t cap, int32_t* n);
/* copy an internal device buffer to host by name (parity tests); returns element count in *n.
 * names: see dsact_debug_names(). */
int dsact_debug_read(dsact_handle* h, const char* name, float* out, size_t cap, size_t* n);
const char* dsact_debug_names(void);
/* Test hooks for the in-launch producer/consumer hand-overs of the merged launches (no reference counterpart: the
 * reference's update is one synchronous CPU call, dsac_v2.py:102-105). A consumer workgroup waits a BOUNDED time for
 * its producers; when it gives up, the next entry point of this library fails with DSACT_E_HIP, the merged launches are
 * disabled for the handle and a captured graph is captured again without them.
 *   dsact_debug_set(h, "withhold_flag", 1|2)    1: one forward producer never raises its ready flag; 2: one slice of the
 *                                               policy backward never arrives (both force the timeout path)
 *   dsact_debug_set(h, "poison_handover", v)    fills every buffer handed from producers to consumers with v (e.g. NaN)
 *   dsact_debug_set(h, "fwd_merge", 0|1)        merged forward launch off / on (when the shape allows it)
 *   dsact_debug_set(h, "pi_merge", 0|1)         policy weight-gradient tiles inside the policy-backward launch off / on
 *   dsact_debug_get(h, "fwd_merge" | "pi_merge" | "fat" | "handoff_failures" | "graph_steps" | "pipe_graph" |
 *                      "act_launch_us" | "act_wait_us", &v) */
int dsact_debug_set(dsact_handle* h, const char* name, double value);
int dsact_debug_get(const dsact_handle* h, const char* name, double* value);
/* stand-alone fused-MLP forward of the policy net on a host batch (sampler / evaluator feed):
 * logits[n*2A] = (mean | std) exactly as StochaPolicy.forward returns (networks/mlp.py:79-100) */
int dsact_policy_forward(dsact_handle* h, const float* obs_host, int32_t n, float* logits_host);
/* OffSampler.sample()'s device work of one environment step in ONE call (training/off_sampler.py:46-54): policy(obs) on the
 * live weights + TanhGaussDistribution.sample() (utils/act_distribution_cls.py:32-42) in the output layer's epilogue, with
 * the caller's standard-normal draw eps[A] (torch.randn(1, A): it consumes the torch generator exactly as Normal.sample()
 * does, and mean + std * eps is bit for bit what Normal.sample() returns). action[A], logp[1] on the host; MLP policies
 * (DSACT_E_INVALID otherwise -- the caller then takes dsact_policy_forward and samples itself). Synchronous. */
int dsact_act_sample(dsact_handle* h, const float* obs_host, const float* eps_host, float* action_host, float* logp_host);

#ifdef __cplusplus
}
#endif
#endif /* DSACT_H */
