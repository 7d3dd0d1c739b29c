// dsact_api.hip -- host side of libdsact.so: handle, HBM layout, task tables, launch sequence,
// hipGraph capture and the extern "C" entry points declared in include/dsact.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <chrono>
#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/dsact.h"
#include "dsact_kernels.h"
#include "dsact_chain.h"
#include "dsact_fat.h"
#include "dsact_act.h"
#include "dsact_host_act.h"
#include "dsact_conv.h"

// every listed template-kernel instantiation is compiled in its family's translation unit (csrc/dsact_tu_<group>.hip, dsact_tu.h)
#define DSACT_KERNEL(group, ...) extern template __global__ void __VA_ARGS__;
#include "dsact_instances.inc"
#undef DSACT_KERNEL

using namespace dsact;

namespace {

enum Chain { C_PI = 0, C_PIT, C_Q1C, C_Q2C, C_Q1T, C_Q2T, C_Q1P, C_Q2P, N_CHAIN };
static const char* kChainName[N_CHAIN] = {"pi", "pit", "q1c", "q2c", "q1t", "q2t", "q1p", "q2p"};
enum Net { N_Q1 = 0, N_Q2, N_POL, N_Q1T, N_Q2T, N_POLT, N_NET };
static const int kChainNet[N_CHAIN] = {N_POL, N_POLT, N_Q1, N_Q2, N_Q1T, N_Q2T, N_Q1, N_Q2};
// chains that are differentiated and their slot in the dZ storage
static const int kDzSlot[N_CHAIN] = {0, -1, 1, 2, -1, -1, 3, 4};

constexpr int kMaxLin = DSACT_MAX_HIDDEN_LAYERS + 1;
constexpr int kActRows = 64;  // rows of the stand-alone policy forward (sampler feed)

// One approximator inside its arena. nblk == 2: the CNN nets' twin `mean` / `log_std` MLPs laid side
// by side (networks/cnn.py:224-229,447-450): layer 0 is one dense (2*H0 x in) matrix, hidden layers are
// two (H x H) blocks, the output layer is the (n_out x 2H) matrix [[w_mean,0],[0,w_ls]]; rows of every
// activation are [mean trunk | log_std trunk]. in[] / out[] are full ROW widths.
struct NetDesc {
  int n_lin = 0;
  int nblk = 1;
  int in[kMaxLin], out[kMaxLin];
  size_t w_off[kMaxLin], b_off[kMaxLin];
  int n_conv = 0;
  size_t cw_off[kMaxConv], cb_off[kMaxConv];
  size_t count = 0;
};

// conv stacks: S_* = (net, image) pairs; the first three are differentiated
enum ConvStack { S_Q1 = 0, S_Q2, S_PI, S_Q1T, S_Q2T, S_PIT, N_STACK };

struct Stage {            // one k_stage launch: <= kMaxProb GEMM problems in the kernel arguments
  std::string name;
  StageArgs args;
  int n_blocks = 0;
  int max_k = 0;          // longest contraction of the stage -> dynamic LDS
  int ts = -1;            // >0: every problem is clean with K == ts*64 -> specialised kernel; 0: general
  int kind = 0;           // 0 forward (KC x KC, bias+GELU), 1 backward (KC x MC, * GELU'), 2 KC x MC plain store
};

struct ProfRec {
  std::string name;
  hipEvent_t e0, e1;
  int blocks;
};

}  // namespace

constexpr int kChainFlagSlices = 64;                 // slices per unit of a merged forward launch (batch <= 256, >= 4 rows each)
constexpr int kChainFlags = 10 * kChainFlagSlices;  // done[6 units] + zdone[q1c, q2c]; twin trunks: 8 x (group, net) partial-output flags + done[pi, pit]
constexpr int kChainCounters = kChMaxL * 8 * kArriveStride;   // behind the flags (+128 ints of padding): per-layer arrival counters (8 replicas each) of the merged policy backward
constexpr int kChainFlagInts = kChainFlags + 128 + kChainCounters;

struct dsact_handle {
  dsact_config cfg;
  int device = 0;
  char err[512] = {0};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t aux_stream = nullptr;   // forked branch: critics' dW + Adam run beside the actor backward chain
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool use_fork = false;
  // conv backward on two queues: the data-gradient chain (dCol -> col2im / direct dX, layer by layer) on the handle's
  // stream, the weight-gradient launches of the same layers on aux_stream -- they only need dY[j], which is ready when the
  // data-gradient launch of layer j starts. ev_conv[j]: dY[j] complete.
  bool conv_fork = false;
  hipEvent_t ev_conv[kMaxConv + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // nets
  int nq = 2;            // critics: 2 (DSAC_V2) or 1 (DSAC_V1)
  NetDesc qd, pd;
  size_t n_q = 0, n_pi = 0, n_online = 0, n_target = 0;
  float *online = nullptr, *target = nullptr, *adam_m = nullptr, *adam_v = nullptr, *grads = nullptr;
  // dims (O: floats of one replay observation; F: observation part of an MLP input row -- == O for the
  // MLP nets, the flattened conv features for the CNN nets)
  int O = 0, F = 0, A = 0, L = 0, B = 0, ldx = 0;
  // CNN encoders
  bool cnn = false;
  int n_conv = 0, Brows = 0;            // Brows = max(B, kActRows): rows the shared geometry tables cover
  ConvGeom cg[kMaxConv];
  int cP = 1;                           // pixels of the last conv layer
  float* img[2];                        // staged minibatch images (obs, obs2), pixel-major
  float* cact[N_STACK][kMaxConv];
  float* cdy[3][kMaxConv];
  float* dcol[3];
  float* dfeat[3];
  float* dwpart[3][kMaxConv];           // weight-gradient partials per differentiated stack and layer
  float *aimg, *aact[kMaxConv];         // stand-alone policy forward
  float* stage_img = nullptr;           // device staging of a host minibatch's images (dsact_load_batch)
  int* idx_iota = nullptr;
  float* Xc[8];                         // MLP input rows per chain
  int w[DSACT_MAX_HIDDEN_LAYERS];     // activation row widths: the wider of the two families per layer (buffer sizes)
  int Lq = 0, Lp = 0;   // hidden layers of the critics / of the policy nets (L = the larger; they differ only with policy_n_hidden)
  int wq[DSACT_MAX_HIDDEN_LAYERS], wp[DSACT_MAX_HIDDEN_LAYERS];   // ... of the critics / of the policy nets (equal unless policy_hidden is set)
  // workspace
  char* ws = nullptr;
  size_t ws_bytes = 0;
  float *X0, *XP, *X2, *rew, *done;
  float *eps_new, *eps_2, *z5, *z6;
  float* Hb[N_CHAIN][DSACT_MAX_HIDDEN_LAYERS];
  float* Gb[N_CHAIN][DSACT_MAX_HIDDEN_LAYERS];
  float* dZ[5][DSACT_MAX_HIDDEN_LAYERS];
  float *logits_pi, *logits_pit, *logp_new, *logp2;
  float *qout_c[2], *qout_t[2], *qout_p[2], *qstd_c[2];
  float* W1p[4];  // zero-padded first-layer weights of q1, q2, q1_target, q2_target  [w0 x ldx]
  bool use_w1p = true;   // false for very wide first layers: copying them every step costs more than unaligned rows
  float* dout[4];
  float *dout_pi, *d_new_act;
  float* qdmean[2] = {nullptr, nullptr}; float* pi_dact = nullptr;   // GELU output layers: the heads' stored d y / d z (carve)
  float* W1aT[2];  // transposed zero-padded action columns of q1 / q2's first layer  [32][w0]
  float *part_loss, *part_heads, *stats, *ones, *std_sums;
  long long* timeline;  // [512][8] stamps of the stage named by DSACT_TIMELINE_STAGE (instrumented builds)
  float *act_scale, *act_center;
  DevState* st = nullptr;
  int* idx_eager = nullptr;
  int* idx_table = nullptr;
  int idx_rows = 0;
  int n_loss_wg = 0, loss_rows = 0, n_heads_wg = 0;
  // stand-alone policy forward
  float *Xact, *Hact[DSACT_MAX_HIDDEN_LAYERS], *Gact, *act_out;
  // tasks
  GemmProb* d_tiles = nullptr;   // per-tile table of the weight/bias-gradient tiles: [q1 | q2 | policy]
  int n_dw_tiles = 0;
  int dw_off[4] = {0, 0, 0, 0};  // start of q1, q2, policy tiles, end
  int dw_pol_rest = 0;           // the policy's OUTPUT-layer tiles come first in its range ([dw_off[2], dw_pol_rest)): they
                                 // only need heads_bwd's results and may ride along in the policy-backward launches
  // split-K weight gradients at batch > 448: dw_chunks chunks of 256 samples, one partial gradient arena each
  int dw_chunks = 1;
  float* dw_parts = nullptr;     // [dw_chunks][dw_part_stride]
  size_t dw_part_stride = 0;
  std::vector<Stage> fwd1, fwd2, bwdq, bwdq_critic, bwdpi, actf;
  Stage dfeat_q, dfeat_pi, dfeat_all;
  // Second batch set (MLP nets): graph replays stage update s+1's minibatch while update s still reads its own
  // (the gather rides in the loss launch, see RideArgs). `alt` holds the set that is NOT selected; select_set()
  // swaps the two, so every enqueue function keeps reading h->X0, h->fwd1, h->d_tiles ... of the selected set.
  struct BatchSet {
    float *X0 = nullptr, *XP = nullptr, *X2 = nullptr, *rew = nullptr, *done = nullptr;
    float *eps_new = nullptr, *eps_2 = nullptr, *z5 = nullptr, *z6 = nullptr;
    float* Xc[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float* W1aT[2] = {nullptr, nullptr};
    std::vector<Stage> fwd1, fwd2;
    GemmProb* d_tiles = nullptr;
  } alt;
  char* alt_ws = nullptr;
  int cur_set = 0;             // 0 outside of graph capture
  // environment switches, read once at dsact_create (getenv walks the whole environment: ~20 calls per eager update
  // were host time on the launch path)
  std::string env_timeline_stage;   // DSACT_TIMELINE_STAGE
  bool env_no_merged_gather = false, env_no_adam_pack = false;
  int n_cu = 256;                       // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int env_conv_dw_nkt = 0;
  int conv_dw_nkt_l[kMaxConv] = {1, 1, 1, 1, 1, 1};   // k-tiles per k_conv_dw workgroup, per layer (DSACT_CONV_DW_NKT_L=a,b,..; DSACT_CONV_DW_NKT: all)
  int env_conv_fwd64_min = 256;         // DSACT_CONV_FWD64_MIN: fewest 64 x 64 tiles a conv forward launch must have to use them
  bool env_no_dcol_ident = false;       // DSACT_NO_DCOL_IDENT: keep dCol + col2im on the layer whose col2im is the identity (round 6 fuses the mask into the product)
  int env_dcol64_min_m = 256;           // DSACT_DCOL64_MIN_M: fewest rows of a dCol product for the 64 x 64 stage tiles (layer 5 at batch 256: 11.4 -> 8.9 us)
  bool env_no_conv_fwd32x64 = false;    // DSACT_NO_CONV_FWD32X64
  int env_conv_dw_reg = 0;              // DSACT_CONV_DW_REG: bit mask of the (narrow) layers whose weight gradient runs on register tiles (k_conv_dw_reg)
  int env_conv_dw_reg_wgs = 512;        // DSACT_CONV_DW_REG_WGS: workgroups (4 waves each) such a launch aims for
  bool env_conv_dw_sb3 = false;         // DSACT_CONV_DW_SB3: layers with three k-tiles per workgroup run the single-buffered form
  bool env_no_conv_fwd64 = false;       // DSACT_NO_CONV_FWD64: wide conv layers' forward on the 32 x 32 tile kernel
  bool env_no_conv_narrow9 = false;     // DSACT_NO_CONV_NARROW9: type_2's third conv layer stays on the LDS-tile forward kernel
  bool mirror_w0 = false;      // set while the merged-gather graph is being captured (see FusedOpt::mir_*)
  bool merged_graph = false;   // the captured graph uses the merged-gather flow
  bool have_local_tail = false;   // the gradient arena's tail holds mean_std of a gradient computed HERE and not yet committed
  long long dev_it_next = -1;  // host shadow of DevState::it_next after graph replays (-1: unknown, upload it)
  hipEvent_t tev0 = nullptr, tev1 = nullptr;   // dsact_time_steps' events, created once
  hipEvent_t uev0 = nullptr, uev1 = nullptr;   // around the last eager update (dsact_step / dsact_compute_grads)
  bool uev_valid = false;
  // replay ring
  long long cap = 0, ptr = 0, size = 0;
  float *rb_obs = nullptr, *rb_obs2 = nullptr, *rb_act = nullptr, *rb_rew = nullptr, *rb_done = nullptr, *rb_logp = nullptr;
  float* stage_dev = nullptr;  // device staging for ring writes
  size_t stage_rows = 0;
  float* stage_pin[2] = {nullptr, nullptr};   // pinned host staging, two slots: dsact_buffer_add returns without a sync
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  unsigned stage_k = 0;
  float* logp_stage = nullptr;                // [B] gathered logp for dsact_read_batch
  // pinned host staging
  int* h_idx[8];
  hipEvent_t h_idx_ev[8];
  int h_idx_slot = 0;
  // rng
  uint64_t rng_seed = 0;
  bool have_batch = false;
  bool limits_set = false;
  // graph
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int graph_steps = 0;
  uint32_t graph_flags = 0;
  // dsact_run_group: captured graphs of OTHER lengths / noise modes than the active one stay instantiated (a trainer's
  // groups between two sampler calls come in a few lengths: sample_interval, and what log / eval / save iterations cut off)
  struct GraphSet {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    hipGraph_t pgraph[4] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t pexec[4] = {nullptr, nullptr, nullptr, nullptr};
    void* pargs[4] = {nullptr, nullptr, nullptr, nullptr};
    bool pipe = false, merged = false, noise_table = false;
    int steps = 0; uint32_t flags = 0;
  };
  std::vector<GraphSet> graph_cache;
  bool graph_noise_table = false;       // the ACTIVE graph's gathers read the noise table (strict RNG) instead of drawing Philox noise
  bool noise_table_on = false;          // set while such a graph is being captured (noise_args)
  bool build_keeps_cache = false;       // set by dsact_run_group around dsact_graph_build: the previous graphs are stashed, not destroyed
  float* noise_table = nullptr;         // [idx_rows][2*B*A + 2*B]: eps_new | eps_2 | z5 | z6 per replayed update
  char* grp_pin[4] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging of dsact_run_group (index rows + noise rows), 4 slots
  size_t grp_pin_bytes = 0;
  hipEvent_t grp_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned grp_k = 0;
  int want_graph_steps = 0; uint32_t want_graph_flags = 0;   // a hand-over failure could not re-capture the graph: built again lazily
  // profiling
  bool profiling = false;
  std::vector<ProfRec> prof;
  // row-slice fused chains (dsact_chain.h): MLP nets, equal hidden widths of 64 / 128 / 256, batch % 16 == 0
  bool unequal_widths = false;          // policy_hidden differs from hidden: tile-stage kernels only
  bool chain_ok = false;
  int cW = 0, cNT = 0;                  // hidden width, W / 64
  int s_obs = 0, s_act = 0, SoT = 0;    // stream steps (4 k each): observation / action segment of a first layer, policy outputs (2A)
  // throughput-regime kernels (dsact_fat.h): 16x16x4 MFMA, 16/32-row slices, style-16 packs. Chosen per direction: the
  // forward launches win from batch 1024 on (33 vs 58 us for group A), the backward ones -- fewer units per launch,
  // hence fewer workgroups -- only from batch 4096 on (measured, profiles/r03_fat_batches.txt)
  bool fat = false, fat_bwd = false;
  int c_obs = 0, c_act = 0, c_out = 0;  // fat mode: 16-k chunks of the observation / action segment of a first layer, of the policy outputs (2A)
  int env_fat_rt = 0;                   // DSACT_FAT_RT=1|2: force the rows per fat workgroup (16 / 32)
  int cRG = 2;                          // row groups of 4 per chain workgroup (8 rows) when 4-row workgroups would oversubscribe the CUs
  int env_chain_rg = 0;                 // DSACT_CHAIN_RG=1|2|4: force
  bool env_no_fat_stage = false;        // DSACT_NO_FAT_STAGE: the throughput-regime forward reads its first layer's rows from global memory
  bool rg4_ok = false;                  // 16-row chain workgroups fit (LDS) and divide the batch
  int n_slices = 0;
  char* pk_ws = nullptr;                // the fragment-major copies
  float* pk_fwd[N_NET][kMaxLin];        // forward copies per net and layer (index L: output layer)
  float* pk_bwd[3][kMaxLin];            // q1, q2, policy: W_l^T for l >= 1 (policy: index L = Wout^T)
  float* pk_w1at[2];                    // q1, q2: (W0[:, F:])^T
  MirrorDesc* d_mir = nullptr;          // [3 nets][L+1]: what the Adam tiles of each weight tensor refresh
  PackJob* d_pack = nullptr; int n_pack_jobs = 0, pack_blocks = 0;
  AdamPackJob* d_apjobs = nullptr;      // k_adam_pack job table (data-parallel graph)
  int n_apjobs = 0, ap_blocks = 0;
  int* chain_flags = nullptr;           // ready flags of the merged forward launch (+ the merged backward's arrival counters behind them)
  bool flags_dirty = false;             // a merged forward was enqueued and nothing has cleared its ready flags since
  // in-launch hand-overs: a consumer that gives up waiting (bounded spin) writes this word, which lives in mapped pinned
  // HOST memory -- every entry point checks it without a device sync and fails the call (check_handoff)
  int* handoff_host = nullptr; int* handoff_dev = nullptr;
  bool in_handoff = false;
  int handoff_failures = 0;
  bool state_invalid = false;           // sticky after a hand-over timeout: the fused Adam / Polyak epilogues ran on whatever the
                                        // consumers found, so parameters / moments / targets are suspect. Every update entry point
                                        // fails (DSACT_E_STATE) until the caller acknowledges with dsact_set_state or
                                        // dsact_bind_arenas -- which it calls after restoring a checkpoint
  int debug_withhold = 0;               // dsact_debug_set("withhold_flag"): tests force the timeout path
  // single-launch acting forward (dsact_act.h): mapped host block = [hand-off word | done counter | logits], device scratch
  unsigned long long* act_out_host = nullptr; unsigned long long* act_out_dev = nullptr;   // 64 (value, call) pairs
  unsigned long long* act_h = nullptr;  // device: [kActMaxLayers][kMaxWidth] (value, call) pairs
  int act_call = 0;
  double act_launch_us = 0.0, act_wait_us = 0.0;   // host time of the last fast acting forward: launch call, completion spin
  bool env_no_fast_act = false;         // DSACT_NO_FAST_ACT: the sampler's forward through the copy + tile-stage path (A/B)
  // host-side acting (dsact_host_act.h): pinned snapshot of the policy net, refreshed on the handle's stream behind every
  // enqueued update that moves the policy; dsact_act_sample / dsact_policy_forward(n = 1) run on the calling thread
  bool host_act = true;                 // DSACT_NO_HOST_ACT / dsact_debug_set("host_act", 0): the one-launch GPU forward instead
  float* pol_host = nullptr;            // pinned: the policy net's n_pi floats in arena order
  hipEvent_t pol_ev = nullptr;
  bool pol_pending = false;             // a copy has been enqueued and its event not yet seen complete
  unsigned long long pol_epoch = 1;     // bumped by every enqueued operation that may change the policy's parameters
  unsigned long long pol_copied = 0;    // epoch the snapshot (or the copy in flight) belongs to
  float* act_buf = nullptr;             // host scratch: two activation buffers + the output layer's products
  hostact::Pool* act_pool = nullptr;    // fork-join helpers for the wide layers (nullptr: the calling thread alone)
  std::vector<int> act_pin;             // CPUs the helpers are pinned to (cores sharing the last-level cache with the caller)
  int act_cpu = -1;                     // CPU of the calling thread when the helpers were (re)pinned
  unsigned act_repins = 0;
  int act_threads = 0;                  // 0: not calibrated yet; DSACT_HOST_ACT_THREADS forces a count
  float act_scale_h[32] = {0}, act_center_h[32] = {0};   // host copies of act_scale / act_center (act_dim <= 32 on this path)
  double act_host_us = 0.0, act_copy_wait_us = 0.0;
  unsigned long long act_host_calls = 0, act_copies = 0;
  bool env_no_conv_dx_mfma = false;     // DSACT_NO_CONV_DX_MFMA: the 16-channel layer's data gradient with k_conv_dx_block (A/B)
  bool fwd_merge = false;               // launches A and B as one (batch <= 256)
  bool pi_merge = false;                // the policy's weight-gradient tiles + the closing block inside the policy-backward launch (batch <= 512; measured equal-to-slower at 1024)
  float* zobs[4];                       // first-layer accumulators after the observation part: q1, q2 (obs), q1_t, q2_t (obs2)
  float* dAq[4];                        // dL/d new_act through q1 / q2  [B][32] (twin trunks: [2 + i] = the log_std trunks' share)
  // twin-trunk nets on the chains (CNN approximators: mean / log_std MLPs over one conv feature row, networks/cnn.py:214-240,
  // 437-461): each trunk is a chain unit of its own over its half of the twin-width buffers (activation packs: features
  // [t*H, (t+1)*H) = floats t*H*B onward; weight packs: tiles / chunks of the trunk's rows / columns)
  bool twin = false;
  float* X0t_net[3] = {nullptr, nullptr, nullptr};   // every net has its own input rows (features): transposed packs for q1, q2, policy
  float* dz0row[3] = {nullptr, nullptr, nullptr};    // row-major dZ[0] of q1c, q2c, pi [B][w[0]]: operand of the dL/d features product
  float* twin_part[8];                               // [B][64] output-layer partials of the first trunks (group x net), parallel trunks
  bool twin_par = false;                             // the two trunks of a net as workgroups of their own (flags: chain_flags)
  bool env_twin_seq = false;                         // DSACT_TWIN_SEQ: one workgroup runs both trunks back to back
  bool twin_merged = false;                          // groups A and B in one launch (table 0), DSACT_TWIN_NO_MERGE: two
  PipeFwd* d_fwdt[2] = {nullptr, nullptr};           // k_chain_fwdt tables (group A, group B), built by dsact_bind_arenas
  PipeFwd* fwdt_host[2] = {nullptr, nullptr};
  float* doutT[3];                      // transposed packs of dL/d(out): q1, q2 [32 x B], policy [roundup32(2A) x B]
  float* X0t = nullptr;                 // transposed pack of the staged minibatch [roundup32(F+A) x B]
  int dw2_off[4] = {0, 0, 0, 0};        // tile ranges of q1, q2, policy in the dw2 problem list
  int dw2_mid = 0;                      // twin trunks: the critics' FIRST-layer tiles are [dw2_mid, dw2_off[2]) (behind their other tiles)
  int n_heads_parts = 0;                // partial (tanh, sigma) sums the last forward wrote
  // pipelined graph (delayed-update-aware software pipelining, k_chain_fwdp): per-minibatch buffers in kPipeSets copies
  // (set 0 = the workspace's own), one captured graph per phase first_iteration % delay_update
  struct PipeSet {
    float *X0 = nullptr, *XP = nullptr, *X2 = nullptr, *rew = nullptr, *done = nullptr;
    float *eps_new = nullptr, *eps_2 = nullptr, *z5 = nullptr, *z6 = nullptr;
    float *logits_pi = nullptr, *logits_pit = nullptr, *logp_new = nullptr, *logp2 = nullptr;
    float* Hpi[DSACT_MAX_HIDDEN_LAYERS]; float* Gpi[DSACT_MAX_HIDDEN_LAYERS];
    float* qout_t[2] = {nullptr, nullptr};
    float* part_heads = nullptr;
    float* X0t = nullptr;
  };
  static constexpr int kPipeSets = 4;
  static constexpr int kPipePhases = 4;
  PipeSet pset[kPipeSets];
  char* pipe_ws = nullptr;
  unsigned long long* pipe_hand[3] = {nullptr, nullptr, nullptr};   // tagged hand-over buffers [B][32] (value, tag): new_act, act2, act2 of the next minibatch
  bool env_no_pipe_tagged = false;      // DSACT_NO_PIPE_TAGGED: ready flags + separate data instead of (value, tag) pairs (A/B)
  bool pipe_defer_now = false;          // set while the update being enqueued defers its (discarded) policy backward
  // merged critic backward + critic tiles + close (k_chain_bwd_qt) on the updates that defer their policy backward
  int* bqt_cnt = nullptr;               // arrival counters [2 critics][8 x kArriveStride], zeroed by the forward launch's bookkeeping block
  int* bqt_tab = nullptr; int bqt_tab_blocks = 0;   // block -> tile table (classes of layers in arrival order, dealt to the XCDs)
  bool env_no_bqt = false;              // DSACT_NO_BQT_MERGE: critics' backward and their tiles stay two launches (A/B)
  bool bqt_now = false;                 // set while such an update is being enqueued
  // k_chain_bwd_qpt: the whole backward of a policy-moving update of the pipelined graph as one launch
  unsigned long long* bqp_pairs[2] = {nullptr, nullptr};   // dL/d new_act through q1 / q2 as (value, tag) pairs [B][32]
  bool env_no_bqp = false;              // DSACT_NO_BQP_MERGE: critics' backward and policy backward stay two launches on those updates (A/B)
  bool bqp_now = false;
  bool pipe_graph = false;              // the captured graphs are the pipelined ones (pgraph / pexec, one per phase)
  hipGraph_t pgraph[kPipePhases] = {nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t pexec[kPipePhases] = {nullptr, nullptr, nullptr, nullptr};
  PipeFwd* pargs[kPipePhases] = {nullptr, nullptr, nullptr, nullptr};   // device: one PipeFwd per captured forward launch
  bool env_no_pipe = false;             // DSACT_NO_PIPE: graph replays without the pipelining (A/B)
  int env_pk_pad = 0;                   // DSACT_PK_PAD: see build_chain
  bool env_no_pipe_warm = true;         // DSACT_PIPE_WARM=1: L2 warm-up touches in the pipelined forward launches (measured: slower, 60.4 vs 59.6 us)
  bool env_no_pipe_defer = false;       // DSACT_NO_PIPE_DEFER: the discarded policy backward stays in its own update's last launch (A/B)
  std::string env_pipe_map;             // DSACT_PIPE_MAP: XCD lists per unit (experiments), see pipe_xcds
  // native collective (RCCL): communicator of this rank, see dsact_comm_init
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  // strict DP
  bool use_std_sums = false;
  bool auto_std_sums = false;
  float* std_sums_own = nullptr;
};

namespace {


// ---- librccl, opened at run time (no link-time dependency: the library also serves single-GPU users) ----------
struct NcclUid { char internal[128]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUid*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
constexpr int kNcclFloat = 7, kNcclSum = 0, kNcclAvg = 4;

const char* rccl_load(const char* path) {
  if (g_rccl.lib) return nullptr;
  const char* cand[3] = {path && path[0] ? path : "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (int i = 0; i < 3 && !lib; ++i) lib = dlopen(cand[i], RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return "librccl.so not found";
  g_rccl.GetUniqueId = (int (*)(NcclUid*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, NcclUid, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) return "librccl.so lacks the nccl* entry points";
  g_rccl.lib = lib;
  return nullptr;
}
const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error"; }

int fail(dsact_handle* h, int code, const char* fmt, ...) {
  if (h) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(h->err, sizeof(h->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(h, DSACT_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

void build_net(NetDesc& d, int in0, const int* hidden, int L, int n_out, int nblk, int n_conv, const ConvGeom* cg) {
  d.n_lin = L + 1;
  d.nblk = nblk;
  d.n_conv = n_conv;
  size_t off = 0;
  for (int j = 0; j < n_conv; ++j) {
    d.cw_off[j] = off; off += (size_t)cg[j].Cout * cg[j].K;
    d.cb_off[j] = off; off += cg[j].Cout;
  }
  int in = in0;
  for (int l = 0; l <= L; ++l) {
    const int out = l < L ? nblk * hidden[l] : n_out;
    d.in[l] = in; d.out[l] = out;
    d.w_off[l] = off;
    // hidden layers of twin trunks hold two (H x H) blocks; layer 0 and the output layer are dense
    off += (nblk == 2 && l > 0 && l < L) ? (size_t)2 * (out / 2) * (in / 2) : (size_t)in * out;
    d.b_off[l] = off; off += out;
    in = out;
  }
  d.count = off;
}

int net_act(const dsact_handle* h, int net);
const NetDesc& net_desc(const dsact_handle* h, int net) { return (net == N_POL || net == N_POLT) ? h->pd : h->qd; }
// hidden layers of the net chain `ch` evaluates (value_hidden_sizes and policy_hidden_sizes may differ in length)
int chain_L(const dsact_handle* h, int ch) { return (ch == C_PI || ch == C_PIT) ? h->Lp : h->Lq; }

// any net with a hidden activation other than GELU: the forward chains run their generic-activation instantiations
// (the generic-activation instantiations also carry the heads' OUTPUT activations, round 6)
bool generic_act(const dsact_handle* h) {
  return h->cfg.value_act != ACT_GELU || h->cfg.policy_act != ACT_GELU || h->cfg.value_out_act != 0 || h->cfg.policy_out_act != 0;
}
// hidden activation of a net's MLP layers (value_hidden_activation / policy_hidden_activation, common_utils.py:16-45)
int net_act(const dsact_handle* h, int net) { return (net == N_POL || net == N_POLT) ? h->cfg.policy_act : h->cfg.value_act; }

// base pointer of a net's parameters inside its arena (online or target)
float* net_base(const dsact_handle* h, int net, float* online, float* target) {
  const size_t q2 = h->nq == 2 ? h->n_q : 0;   // DSAC_V1 has one critic: the q2 slots alias q1 (never written)
  switch (net) {
    case N_Q1: return online;
    case N_Q2: return online + q2;
    case N_POL: return online + h->nq * h->n_q;
    case N_Q1T: return target;
    case N_Q2T: return target + q2;
    default: return target + h->nq * h->n_q;
  }
}
float* net_params(const dsact_handle* h, int net) { return net_base(h, net, h->online, h->target); }
float* net_grads(const dsact_handle* h, int net) { return net_base(h, net, h->grads, nullptr); }

template <typename... KArgs, typename... Args>
int launch_on(dsact_handle* h, hipStream_t stream, const char* name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, args...);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, DSACT_E_HIP, "launch %s failed: %s", name, hipGetErrorString(e));
  return DSACT_OK;
}

template <typename... KArgs, typename... Args>
int launch(dsact_handle* h, const char* name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
  if (h->profiling) {
    ProfRec r;
    r.name = name;
    r.blocks = (int)(grid.x * grid.y * grid.z);
    HIPCHK(h, hipEventCreate(&r.e0));
    HIPCHK(h, hipEventCreate(&r.e1));
    // start / stop events attached to the dispatch itself: the kernel's own begin and end timestamps (what rocprofv3
    // reports as its duration), not the stream time between two event-record commands
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, h->stream, r.e0, r.e1, 0, args...);
    h->prof.push_back(r);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, shmem, h->stream, args...);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, DSACT_E_HIP, "launch %s failed: %s", name, hipGetErrorString(e));
  return DSACT_OK;
}

#define TRY(x)            \
  do {                    \
    int rc_ = (x);        \
    if (rc_ != DSACT_OK) return rc_; \
  } while (0)

// pixels per split of a conv weight-gradient contraction (multiple of 64): the SMALLEST the launch may use -- the partial
// buffers are sized for it
size_t conv_dw_chunk(size_t M) { (void)M; return 256; }   // (round 3: 1024 from 64 Ki pixels on; the register-tile launches want finer chunks)

// ---- workspace carving ------------------------------------------------------------------------
struct Carver {
  size_t off = 0;
  char* base = nullptr;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

void carve(dsact_handle* h, Carver& c) {
  const size_t B = h->B;
  const int A = h->A, L = h->L;
  h->st = c.take<DevState>(1);
  h->X0 = c.take<float>(B * h->ldx);
  h->XP = c.take<float>(B * h->ldx);
  h->X2 = c.take<float>(B * h->ldx);
  if (!h->cnn) {
    // MLP nets read the observation itself: chains on the same (obs, action) source share one row buffer
    h->Xc[C_PI] = h->Xc[C_Q1C] = h->Xc[C_Q2C] = h->X0;
    h->Xc[C_PIT] = h->Xc[C_Q1T] = h->Xc[C_Q2T] = h->X2;
    h->Xc[C_Q1P] = h->Xc[C_Q2P] = h->XP;
  } else {
    // CNN nets: every net has its own encoder, so every chain has its own [features | action] rows
    h->Xc[C_Q1C] = h->X0; h->Xc[C_Q1P] = h->XP; h->Xc[C_Q1T] = h->X2;
    for (int ch : {C_PI, C_PIT, C_Q2C, C_Q2T, C_Q2P}) h->Xc[ch] = c.take<float>(B * h->ldx);
    const size_t R = h->Brows;
    h->img[0] = c.take<float>(B * h->O);
    h->img[1] = c.take<float>(B * h->O);
    size_t dcol_max = 4;
    for (int j = 0; j < h->n_conv; ++j) {
      const ConvGeom& g = h->cg[j];
      const size_t M = B * g.OH * g.OW;
      for (int st = 0; st < N_STACK; ++st) h->cact[st][j] = c.take<float>(M * g.Cout);
      for (int st = 0; st < 3; ++st) h->cdy[st][j] = c.take<float>(M * g.Cout);
      if (j > 0 && M * g.K > dcol_max) dcol_max = M * g.K;
      const size_t chunks = (M + conv_dw_chunk(M) - 1) / conv_dw_chunk(M);
      for (int st = 0; st < 3; ++st) h->dwpart[st][j] = c.take<float>(chunks * g.Cout * (g.K + 4));
      h->aact[j] = c.take<float>((size_t)kActRows * g.OH * g.OW * g.Cout);
    }
    for (int st = 0; st < 3; ++st) {
      h->dcol[st] = c.take<float>(dcol_max);
      h->dfeat[st] = c.take<float>(B * h->F);
    }
    h->aimg = c.take<float>((size_t)kActRows * h->O);
    h->idx_iota = c.take<int>(R);
  }
  h->rew = c.take<float>(B);
  h->done = c.take<float>(B);
  h->eps_new = c.take<float>(B * A);
  h->eps_2 = c.take<float>(B * A);
  h->z5 = c.take<float>(B);
  h->z6 = c.take<float>(B);
  for (int ch = 0; ch < N_CHAIN; ++ch)
    for (int l = 0; l < L; ++l) {
      h->Hb[ch][l] = c.take<float>(B * h->w[l]);
      h->Gb[ch][l] = c.take<float>(B * h->w[l]);
    }
  for (int s = 0; s < 5; ++s)
    for (int l = 0; l < L; ++l) h->dZ[s][l] = c.take<float>(B * h->w[l]);
  h->logits_pi = c.take<float>(B * 2 * A);
  h->logits_pit = c.take<float>(B * 2 * A);
  h->logp_new = c.take<float>(B);
  h->logp2 = c.take<float>(B);
  for (int i = 0; i < 2; ++i) {
    h->qout_c[i] = c.take<float>(B * 2);
    h->qout_t[i] = c.take<float>(B * 2);
    h->qout_p[i] = c.take<float>(B * 2);
    h->qstd_c[i] = c.take<float>(B * 2);
  }
  for (int i = 0; i < 4; ++i) h->dout[i] = c.take<float>(B * 2);
  h->dout_pi = c.take<float>(B * 2 * A);
  h->d_new_act = c.take<float>(B * A);
  // GELU output layers (tile-stage kernels only): d y / d z of the heads that are differentiated later
  for (int i = 0; i < 2; ++i) h->qdmean[i] = h->cfg.value_out_act == OUT_ACT_GELU ? c.take<float>(B) : nullptr;
  h->pi_dact = h->cfg.policy_out_act == OUT_ACT_GELU ? c.take<float>(B * 2 * A) : nullptr;
  h->W1aT[0] = c.take<float>((size_t)32 * h->w[0]);
  h->W1aT[1] = c.take<float>((size_t)32 * h->w[0]);
  h->part_loss = c.take<float>(B * kLossPart);
  for (int i = 0; i < 4; ++i) h->W1p[i] = c.take<float>(h->use_w1p ? (size_t)h->w[0] * h->ldx : 4);
  h->part_heads = c.take<float>((size_t)h->n_heads_wg * 2);
  h->stats = c.take<float>(16 * (1 + DSACT_STATS_SLOTS));   // [0]: dsact_read_stats; [1 + slot]: the snapshot ring
  h->ones = c.take<float>(B);
  h->std_sums = c.take<float>(2);
  h->timeline = c.take<long long>(1024 * 16);
  h->dw_parts = c.take<float>(h->dw_chunks > 1 ? (size_t)h->dw_chunks * h->dw_part_stride : 4);
  for (int i = 0; i < 4; ++i) h->zobs[i] = c.take<float>(B * h->w[0]);
  h->chain_flags = c.take<int>(kChainFlagInts);   // [unit 0..5][slice] ready flags of the merged forward launch, then the spin-timeout word
  for (int i = 0; i < 4; ++i) h->dAq[i] = c.take<float>(B * 32);
  for (int i = 0; i < 2; ++i) h->doutT[i] = c.take<float>(B * 32);
  h->doutT[2] = c.take<float>(B * (size_t)((2 * A + 31) / 32 * 32));
  h->X0t = c.take<float>(B * (size_t)((h->F + A + 31) / 32 * 32));
  if (h->twin) {
    h->X0t_net[0] = h->X0t;
    for (int i = 1; i < 3; ++i) h->X0t_net[i] = c.take<float>(B * (size_t)((h->F + A + 31) / 32 * 32));
    for (int i = 0; i < 3; ++i) h->dz0row[i] = c.take<float>(B * h->w[0]);
    for (int i = 0; i < 8; ++i) h->twin_part[i] = c.take<float>(B * 64);
  }
  h->act_scale = c.take<float>(A);
  h->act_center = c.take<float>(A);
  h->idx_eager = c.take<int>(B);
  h->Xact = c.take<float>((size_t)kActRows * h->ldx);
  for (int l = 0; l < L; ++l) h->Hact[l] = c.take<float>((size_t)kActRows * h->w[l]);
  int wmax = 0;
  for (int l = 0; l < L; ++l) wmax = h->w[l] > wmax ? h->w[l] : wmax;
  h->Gact = c.take<float>((size_t)kActRows * wmax);
  h->act_out = c.take<float>((size_t)kActRows * 2 * A);
}

// ---- stage descriptions ---------------------------------------------------------------------------
const float* chain_input(const dsact_handle* h, int ch) { return h->Xc[ch]; }

int tiles_of(int n, int t) { return (n + t - 1) / t; }
// contraction length of one weight-gradient tile
int dw_k(const dsact_handle* h) { return h->dw_chunks > 1 ? 256 : h->B; }

void stage_add(Stage& s, GemmProb g) {
  g.tiles_n = tiles_of(g.N, TN);
  const int nt = tiles_of(g.M, TM) * g.tiles_n;
  s.n_blocks += nt;
  g.tile_end = s.n_blocks;
  if (g.K > s.max_k) s.max_k = g.K;
  const int clean = (g.M % TM == 0 && g.N % TN == 0 && g.K % BK == 0) ? g.K / BK : 0;
  s.ts = s.ts < 0 ? clean : (s.ts == clean ? clean : 0);
  if (s.args.n_prob >= kMaxProb) abort();  // programming error: a stage never has more than 4 chains x 2 trunks
  s.args.p[s.args.n_prob++] = g;
}

// forward problems of layer l of chain ch: one dense product, or one per trunk for the hidden layers of
// twin-trunk (CNN) nets
void fwd_probs(const dsact_handle* h, int ch, int l, const float* x0, int ldx0, int M, float* const* Hrow, float* Grow,
               std::vector<GemmProb>& out) {
  const int net = kChainNet[ch];
  const NetDesc& d = net_desc(h, net);
  const float* base = net_params(h, net);
  const int nb = (d.nblk == 2 && l > 0) ? 2 : 1;
  const int hb = d.out[l] / nb, kb = d.in[l] / nb;
  for (int b = 0; b < nb; ++b) {
    GemmProb t;
    memset(&t, 0, sizeof(t));
    t.P = l == 0 ? x0 : Hrow[l - 1] + (size_t)b * kb;
    t.ldp = l == 0 ? ldx0 : d.in[l];
    t.Q = base + d.w_off[l] + (size_t)b * hb * kb;
    t.ldq = kb;
    if (l == 0 && net != N_POL && net != N_POLT && h->use_w1p) {
      // Q nets: rows of F+A floats are not 16-byte aligned -> use the zero-padded copy (k_gather repack)
      const int slot = net == N_Q1 ? 0 : net == N_Q2 ? 1 : net == N_Q1T ? 2 : 3;
      t.Q = h->W1p[slot];
      t.ldq = h->ldx;
    }
    t.aux = base + d.b_off[l] + (size_t)b * hb;
    t.C0 = Hrow[l] + (size_t)b * hb;
    t.C1 = Grow + (size_t)b * hb;
    t.ldc = d.out[l];
    t.M = M; t.N = hb; t.K = kb;
    t.act = net_act(h, net);
    out.push_back(t);
  }
}


void select_set(dsact_handle* h, int set) {
  if (set == h->cur_set) return;
  dsact_handle::BatchSet& a = h->alt;
  std::swap(h->X0, a.X0); std::swap(h->XP, a.XP); std::swap(h->X2, a.X2); std::swap(h->rew, a.rew); std::swap(h->done, a.done);
  std::swap(h->eps_new, a.eps_new); std::swap(h->eps_2, a.eps_2); std::swap(h->z5, a.z5); std::swap(h->z6, a.z6);
  for (int i = 0; i < 8; ++i) std::swap(h->Xc[i], a.Xc[i]);
  for (int i = 0; i < 2; ++i) std::swap(h->W1aT[i], a.W1aT[i]);
  h->fwd1.swap(a.fwd1); h->fwd2.swap(a.fwd2);
  std::swap(h->d_tiles, a.d_tiles);
  h->cur_set = set;
}

// the second batch set's buffers (MLP nets only); zeroed like the workspace
int alloc_alt_set(dsact_handle* h) {
  if (h->cnn || h->alt_ws) return DSACT_OK;
  const size_t B = h->B;
  const int A = h->A;
  for (int pass = 0; pass < 2; ++pass) {
    Carver c;
    c.base = pass ? h->alt_ws : nullptr;
    dsact_handle::BatchSet& a = h->alt;
    a.X0 = c.take<float>(B * h->ldx); a.XP = c.take<float>(B * h->ldx); a.X2 = c.take<float>(B * h->ldx);
    a.rew = c.take<float>(B); a.done = c.take<float>(B);
    a.eps_new = c.take<float>(B * A); a.eps_2 = c.take<float>(B * A); a.z5 = c.take<float>(B); a.z6 = c.take<float>(B);
    a.W1aT[0] = c.take<float>((size_t)32 * h->w[0]); a.W1aT[1] = c.take<float>((size_t)32 * h->w[0]);
    if (!pass) {
      HIPCHK(h, hipMalloc((void**)&h->alt_ws, c.off + 256));
      HIPCHK(h, hipMemset(h->alt_ws, 0, c.off + 256));
    }
  }
  dsact_handle::BatchSet& a = h->alt;
  a.Xc[C_PI] = a.Xc[C_Q1C] = a.Xc[C_Q2C] = a.X0;
  a.Xc[C_PIT] = a.Xc[C_Q1T] = a.Xc[C_Q2T] = a.X2;
  a.Xc[C_Q1P] = a.Xc[C_Q2P] = a.XP;
  return DSACT_OK;
}


// ---- row-slice fused chains: packed copies, mirror descriptors, pack jobs -------------------------------------
int roundup(int v, int m) { return (v + m - 1) / m * m; }

int build_chain(dsact_handle* h) {
  if (!h->chain_ok) return DSACT_OK;
  if (h->pk_ws) return DSACT_OK;   // sized by the configuration, not by the arenas: built once
  const int L = h->L, W = h->cW, F = h->F, A = h->A;
  const int tiles = W / 64, SH = W / 4, CH = W / 16;       // style 44: 64-row tiles, k4 steps; style 16: chunks of 16 k
  // fat mode (dsact_fat.h): EVERY pack is style 16 (16-row tiles x chunks of 16 k; a tile-chunk block is 256 floats like
  // a style-44 step, so the sizes below count blocks either way)
  const bool fat = h->fat, fatb = h->fat_bwd;
  const int tiles_f = fat ? W / 16 : tiles, S_hid = fat ? CH : SH + (h->fat ? 0 : h->env_pk_pad);
  const int tiles_b = fatb ? W / 16 : tiles, S_hidb = fatb ? CH : SH, S_out = fatb ? h->c_out : h->SoT;
  const int tpad = fat ? 0 : h->env_pk_pad;   // style-44 forward packs: padding steps behind every tile (DSACT_PK_PAD)
  const int C0q = fat ? h->c_obs + h->c_act : h->s_obs + h->s_act + tpad, C0p = fat ? h->c_obs : h->s_obs + tpad;
  const int nth_q = 1, nth_p = (2 * A + 15) / 16, nta = (A + 15) / 16;
  // twin trunks (nb = 2): a net's first layer is the dense [2W x in] matrix = 2 x tiles row tiles (trunk t: the second half);
  // hidden layers are two [W x W] blocks with a pack each (trunk t: t * tiles * steps * 256 floats further); the output layer
  // is the dense [n_out x 2W] matrix: 2 x CH chunks per 16-row tile (trunk t: chunks [t*CH, (t+1)*CH)); its transpose (policy)
  // has 2 x tiles row tiles; (W0[:, F:])^T is [A x 2W]: 2 x CH chunks per tile
  const int nb = h->twin ? 2 : 1;
  // carve the packed copies
  for (int pass = 0; pass < 2; ++pass) {
    Carver c;
    c.base = pass ? h->pk_ws : nullptr;
    for (int net = 0; net < N_NET; ++net) {
      const bool pol = net == N_POL || net == N_POLT;
      h->pk_fwd[net][0] = c.take<float>((size_t)nb * tiles_f * (pol ? C0p : C0q) * 256);
      for (int l = 1; l < L; ++l) h->pk_fwd[net][l] = c.take<float>((size_t)nb * tiles_f * S_hid * 256);
      h->pk_fwd[net][L] = c.take<float>((size_t)(pol ? nth_p : nth_q) * nb * CH * 256);
    }
    for (int n3 = 0; n3 < 3; ++n3) {
      for (int l = 0; l <= L; ++l) h->pk_bwd[n3][l] = nullptr;
      for (int l = 1; l < L; ++l) h->pk_bwd[n3][l] = c.take<float>((size_t)nb * tiles_b * S_hidb * 256);
    }
    h->pk_bwd[2][L] = c.take<float>((size_t)nb * tiles_b * S_out * 256);
    for (int i = 0; i < 2; ++i) h->pk_w1at[i] = c.take<float>((size_t)nta * nb * CH * 256);
    if (!pass) {
      HIPCHK(h, hipMalloc((void**)&h->pk_ws, c.off + 256));
      HIPCHK(h, hipMemset(h->pk_ws, 0, c.off + 256));   // the zero padding of every copy is written here, once
    }
  }
  // twin trunks: the packed copies are rebuilt from the arenas at the start of every update (the pack blocks ride in the
  // image gather; there is no replayed graph without that launch), so the weight-gradient tiles carry no mirror descriptors
  if (h->twin) return DSACT_OK;
  // what the Adam tiles of (q1, q2, policy) x layer refresh
  const int on3[3] = {N_Q1, N_Q2, N_POL}, tg3[3] = {N_Q1T, N_Q2T, N_POLT};
  std::vector<MirrorDesc> mir((size_t)3 * (L + 1));
  for (int n3 = 0; n3 < 3; ++n3)
    for (int l = 0; l <= L; ++l) {
      MirrorDesc& m = mir[(size_t)n3 * (L + 1) + l];
      memset(&m, 0, sizeof(m));
      const bool pol = n3 == 2;
      m.fwd = h->pk_fwd[on3[n3]][l]; m.fwd_t = h->pk_fwd[tg3[n3]][l];
      m.fwd_44 = (l < L && !fat) ? 1 : 0;                         // the output layers are narrow products (style 16)
      m.fwd_C = l == 0 ? (pol ? C0p : C0q) : (l < L ? S_hid : CH);
      m.F = (l == 0 && !pol) ? F : (1 << 30); m.Fp = fat ? 16 * h->c_obs : 4 * h->s_obs;
      if (l >= 1 && l < L) { m.bwd = h->pk_bwd[n3][l]; m.bwd_44 = fatb ? 0 : 1; m.bwd_C = S_hidb; m.bwd_k0 = 0; }
      if (l == 0 && !pol) { m.bwd = h->pk_w1at[n3]; m.bwd_44 = 0; m.bwd_C = CH; m.bwd_k0 = F; }
      if (l == L && pol) { m.bwd = h->pk_bwd[2][L]; m.bwd_44 = fatb ? 0 : 1; m.bwd_C = S_out; m.bwd_k0 = 0; }
    }
  HIPCHK(h, hipMalloc((void**)&h->d_mir, mir.size() * sizeof(MirrorDesc)));
  HIPCHK(h, hipMemcpy(h->d_mir, mir.data(), mir.size() * sizeof(MirrorDesc), hipMemcpyHostToDevice));
  return DSACT_OK;
}

// jobs of k_pack: every weight tensor of the six nets -> its packed copies (needs the arenas)
int build_pack_jobs(dsact_handle* h) {
  if (!h->chain_ok) return DSACT_OK;
  const int L = h->L;
  std::vector<PackJob> jobs;
  int blocks = 0;
  if (h->twin) {
    // one job per contiguous row-major matrix of the arena: first layer (dense), hidden layer x trunk, output layer (dense)
    const int W = h->cW, tiles = W / 64, SH = W / 4, CH = W / 16;
    for (int net = 0; net < N_NET; ++net) {
      if (h->nq == 1 && (net == N_Q2 || net == N_Q2T)) continue;   // one critic (DSAC_V1)
      const NetDesc& d = net_desc(h, net);
      const bool target = net >= N_Q1T, pol = net == N_POL || net == N_POLT;
      const int n3 = net % 3;
      for (int l = 0; l <= L; ++l)
        for (int t = 0; t < ((l >= 1 && l < L) ? 2 : 1); ++t) {
          PackJob j;
          memset(&j, 0, sizeof(j));
          MirrorDesc& m = j.m;
          m.F = 1 << 30; m.Fp = 4 * h->s_obs;
          if (l == 0) {
            j.src = net_params(h, net) + d.w_off[0]; j.N = d.out[0]; j.K = d.in[0];
            m.fwd = h->pk_fwd[net][0]; m.fwd_44 = 1; m.fwd_C = pol ? h->s_obs : h->s_obs + h->s_act;
            if (!pol) m.F = h->F;
            if (!pol && !target) { m.bwd = h->pk_w1at[n3]; m.bwd_44 = 0; m.bwd_C = 2 * CH; m.bwd_k0 = h->F; }
          } else if (l < L) {
            j.src = net_params(h, net) + d.w_off[l] + (size_t)t * W * W; j.N = W; j.K = W;
            m.fwd = h->pk_fwd[net][l] + (size_t)t * tiles * SH * 256; m.fwd_44 = 1; m.fwd_C = SH;
            if (!target) { m.bwd = h->pk_bwd[n3][l] + (size_t)t * tiles * SH * 256; m.bwd_44 = 1; m.bwd_C = SH; m.bwd_k0 = 0; }
          } else {
            j.src = net_params(h, net) + d.w_off[L]; j.N = d.out[L]; j.K = d.in[L];
            m.fwd = h->pk_fwd[net][L]; m.fwd_44 = 0; m.fwd_C = 2 * CH;
            if (pol && !target) { m.bwd = h->pk_bwd[2][L]; m.bwd_44 = 1; m.bwd_C = h->SoT; m.bwd_k0 = 0; }
          }
          j.is_target = target ? 1 : 0;
          blocks += (j.N + 15) / 16;
          j.block_end = blocks;
          jobs.push_back(j);
        }
    }
  }
  std::vector<MirrorDesc> mir((size_t)3 * (L + 1));
  if (!h->twin) HIPCHK(h, hipMemcpy(mir.data(), h->d_mir, mir.size() * sizeof(MirrorDesc), hipMemcpyDeviceToHost));
  for (int net = 0; net < N_NET && !h->twin; ++net) {
    if (h->nq == 1 && (net == N_Q2 || net == N_Q2T)) continue;   // one critic (DSAC_V1)
    const NetDesc& d = net_desc(h, net);
    const bool target = net >= N_Q1T;
    const int n3 = net % 3;
    for (int l = 0; l <= L; ++l) {
      PackJob j;
      memset(&j, 0, sizeof(j));
      j.src = net_params(h, net) + d.w_off[l]; j.N = d.out[l]; j.K = d.in[l];
      j.m = mir[(size_t)n3 * (L + 1) + l];
      if (target) { j.m.fwd = j.m.fwd_t; j.m.bwd = nullptr; }
      j.m.fwd_t = nullptr;
      j.is_target = target ? 1 : 0;
      blocks += (j.N + 15) / 16;
      j.block_end = blocks;
      jobs.push_back(j);
    }
  }
  if (h->d_pack) { hipFree(h->d_pack); h->d_pack = nullptr; }
  HIPCHK(h, hipMalloc((void**)&h->d_pack, jobs.size() * sizeof(PackJob)));
  HIPCHK(h, hipMemcpy(h->d_pack, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
  h->n_pack_jobs = (int)jobs.size();
  h->pack_blocks = blocks;
  return DSACT_OK;
}

// after_update: the pack follows this update's optimiser pass (data-parallel graph) -- the target nets' copies are
// rebuilt only on delayed-update steps; otherwise (start of an eager step: anything may have written the arenas) all
int enqueue_pack(dsact_handle* h, bool after_update = false) {
  PackArgs a;
  a.jobs = h->d_pack; a.n_jobs = h->n_pack_jobs;
  a.targets_if = after_update ? &h->st->do_delayed : nullptr;
  return launch(h, "pack", k_pack, dim3(h->pack_blocks), dim3(kThreads), 0, a);
}

// job table of k_adam_pack: the weight tensors of the online nets in arena order, 16 rows x 256 columns per block
int build_adam_pack_jobs(dsact_handle* h) {
  const int L = h->L;
  const int chs[3] = {C_Q1C, C_Q2C, C_PI};
  std::vector<AdamPackJob> jobs;
  int blocks = 0;
  for (int n3 = 0; n3 < 3; ++n3) {
    if (n3 == 1 && h->nq == 1) continue;   // one critic (DSAC_V1)
    const int net = kChainNet[chs[n3]];
    const NetDesc& d = net_desc(h, net);
    const long long base = (long long)(net_grads(h, net) - h->grads);
    for (int l = 0; l <= L; ++l) {
      AdamPackJob j;
      memset(&j, 0, sizeof(j));
      j.w_idx = base + (long long)d.w_off[l]; j.b_idx = base + (long long)d.b_off[l];
      j.N = d.out[l]; j.K = d.in[l]; j.col_chunks = (j.K + 63) / 64;
      j.mir = h->d_mir ? h->d_mir + (size_t)n3 * (L + 1) + l : nullptr;
      blocks += ((j.N + 15) / 16) * j.col_chunks;
      j.block_end = blocks;
      jobs.push_back(j);
    }
  }
  if (h->d_apjobs) { hipFree(h->d_apjobs); h->d_apjobs = nullptr; }
  HIPCHK(h, hipMalloc((void**)&h->d_apjobs, jobs.size() * sizeof(AdamPackJob)));
  HIPCHK(h, hipMemcpy(h->d_apjobs, jobs.data(), jobs.size() * sizeof(AdamPackJob), hipMemcpyHostToDevice));
  h->n_apjobs = (int)jobs.size();
  h->ap_blocks = blocks;
  return DSACT_OK;
}

FusedOpt fused_opt(const dsact_handle* h, bool enable);

// Adam / Polyak on the (averaged) gradient arena + the packed copies, one pass; the extra block closes the update
int enqueue_adam_pack(dsact_handle* h) {
  AdamPackArgs a;
  a.jobs = h->d_apjobs; a.n_jobs = h->n_apjobs; a.n_blocks = h->ap_blocks;
  a.fo = fused_opt(h, true);
  h->have_local_tail = false;   // finalize_update commits the tail (the data-parallel graph always computed it here)
  return launch(h, "adam_pack", k_adam_pack, dim3(h->ap_blocks + 1), dim3(kThreads), 0, a);
}

int build_tasks(dsact_handle* h) {
  const int L = h->L, B = h->B;
  auto fresh = [](const std::string& name, int kind) {
    Stage s;
    s.name = name; s.kind = kind; s.n_blocks = 0;
    memset(&s.args, 0, sizeof(s.args));
    return s;
  };
  std::vector<GemmProb> pv;
  // forward group A: policy(obs), policy_target(obs2), q1/q2(obs,act); group B: q1_t/q2_t(obs2,act2), q1/q2(obs,new_act)
  const bool twin = h->nq == 2;
  const std::vector<int> g1 = twin ? std::vector<int>{C_PI, C_PIT, C_Q1C, C_Q2C} : std::vector<int>{C_PI, C_PIT, C_Q1C};
  const std::vector<int> g2 = twin ? std::vector<int>{C_Q1T, C_Q2T, C_Q1P, C_Q2P} : std::vector<int>{C_Q1T, C_Q1P};
  h->fwd1.clear(); h->fwd2.clear();
  for (int grp = 0; grp < 2; ++grp)
    for (int l = 0; l < (grp == 0 ? L : h->Lq); ++l) {   // (group B holds critics only)
      Stage s = fresh(std::string(grp == 0 ? "fwdA_l" : "fwdB_l") + std::to_string(l), 0);
      pv.clear();
      for (size_t i = 0; i < (grp == 0 ? g1 : g2).size(); ++i) {
        const int ch = grp == 0 ? g1[i] : g2[i];
        if (l >= chain_L(h, ch)) continue;   // the shallower family has no layer l
        fwd_probs(h, ch, l, chain_input(h, ch), h->ldx, B, h->Hb[ch], h->Gb[ch][l], pv);
      }
      for (const GemmProb& g : pv) stage_add(s, g);
      (grp == 0 ? h->fwd1 : h->fwd2).push_back(s);
    }
  // backward through hidden layers: dZ[l-1] = (dZ[l] W_l) * G[l-1]   (per trunk for twin nets)
  auto bwd_probs = [&](Stage& s, int ch, int l) {
    const int net = kChainNet[ch];
    const NetDesc& d = net_desc(h, net);
    const float* base = net_params(h, net);
    const int slot = kDzSlot[ch];
    const int nb = d.nblk;
    const int hb = d.out[l] / nb, kb = d.in[l] / nb;
    for (int b = 0; b < nb; ++b) {
      GemmProb t;
      memset(&t, 0, sizeof(t));
      t.P = h->dZ[slot][l] + (size_t)b * hb; t.ldp = d.out[l];
      t.Q = base + d.w_off[l] + (size_t)b * hb * kb; t.ldq = kb;
      t.aux = h->Gb[ch][l - 1] + (size_t)b * kb; t.ldaux = d.in[l];
      t.C0 = h->dZ[slot][l - 1] + (size_t)b * kb; t.ldc = d.in[l];
      t.M = B; t.N = kb; t.K = hb;
      stage_add(s, t);
    }
  };
  h->bwdq.clear(); h->bwdq_critic.clear(); h->bwdpi.clear();
  for (int l = h->Lq - 1; l >= 1; --l) {
    Stage s = fresh("bwdQ_l" + std::to_string(l), 1);
    for (int ch : twin ? std::vector<int>{C_Q1C, C_Q2C, C_Q1P, C_Q2P} : std::vector<int>{C_Q1C, C_Q1P}) bwd_probs(s, ch, l);
    h->bwdq.push_back(s);
    Stage c = fresh("bwdQc_l" + std::to_string(l), 1);  // off iterations of the delayed update: critics only
    for (int ch : twin ? std::vector<int>{C_Q1C, C_Q2C} : std::vector<int>{C_Q1C}) bwd_probs(c, ch, l);
    h->bwdq_critic.push_back(c);
  }
  for (int l = h->Lp - 1; l >= 1; --l) {
    Stage s = fresh("bwdPi_l" + std::to_string(l), 1);
    bwd_probs(s, C_PI, l);
    h->bwdpi.push_back(s);
  }
  // stand-alone policy forward (kActRows rows)
  h->actf.clear();
  for (int l = 0; l < h->Lp; ++l) {
    Stage s = fresh("act_l" + std::to_string(l), 0);
    pv.clear();
    fwd_probs(h, C_PI, l, h->Xact, h->ldx, kActRows, h->Hact, h->Gact, pv);
    for (const GemmProb& g : pv) stage_add(s, g);
    h->actf.push_back(s);
  }
  // CNN nets: gradient w.r.t. the conv features, dFeat = dZ0 . W0[:, :F]  (plain store)
  // (round 6: a conv stack that ends in ONE pixel -- type_2 -- has features == last-layer activations in the same order, so the
  //  ReLU mask of k_feat_bwd is applied in this product's epilogue and the result lands in the last layer's dY: kind 3, one launch less)
  const bool feat_ident = h->cnn && h->cP == 1 && !h->env_no_dcol_ident;
  h->dfeat_q = fresh("dfeat_q", feat_ident ? 3 : 2);
  h->dfeat_pi = fresh("dfeat_pi", feat_ident ? 3 : 2);
  h->dfeat_all = fresh("dfeat", feat_ident ? 3 : 2);
  if (h->cnn) {
    for (int ch : {C_Q1C, C_Q2C, C_PI}) {
      if (ch == C_Q2C && h->nq == 1) continue;   // one critic (DSAC_V1)
      const int net = kChainNet[ch];
      const NetDesc& d = net_desc(h, net);
      GemmProb t;
      memset(&t, 0, sizeof(t));
      t.P = h->dZ[kDzSlot[ch]][0]; t.ldp = d.out[0];
      // chain mode: dZ[0] is a transposed pack; the backward chains also leave it row-major (dz0row), and the product reads
      // the arena's W0 -- it runs before the launch whose tiles update that net
      if (h->twin) { t.P = h->dz0row[ch == C_PI ? 2 : ch - C_Q1C]; t.Q = net_params(h, net) + d.w_off[0]; t.ldq = d.in[0]; }
      else if (ch == C_PI) { t.Q = net_params(h, net) + d.w_off[0]; t.ldq = d.in[0]; }
      else if (h->use_w1p) { t.Q = h->W1p[ch == C_Q1C ? 0 : 1]; t.ldq = h->ldx; }  // this step's pre-update copy
      else { t.Q = net_params(h, net) + d.w_off[0]; t.ldq = d.in[0]; }              // (dfeat_q runs before the critics' update)
      const int stack = ch == C_PI ? h->nq : ch - C_Q1C;               // (stack numbering: q x nq, then the policy)
      t.C0 = h->dfeat[stack]; t.ldc = h->F;
      t.M = B; t.N = h->F; t.K = d.out[0];
      if (feat_ident) { t.C0 = h->cdy[stack][h->n_conv - 1]; t.aux = h->cact[stack][h->n_conv - 1]; t.ldaux = h->F; t.act = MULG_RELU_MASK; }
      stage_add(ch == C_PI ? h->dfeat_pi : h->dfeat_q, t);
      stage_add(h->dfeat_all, t);
    }
  }
  // weight / bias gradients of q1, q2, policy: one table entry per 32x32 tile
  std::vector<GemmProb> tiles;
  // split-K: the batch-long contraction is cut into chunks of 256 samples (the size the tile kernel is tuned for:
  // whole-K register prefetch, one barrier); chunk c accumulates into its own partial arena, k_sum_parts adds them
  const int chunks = h->dw_chunks;
  auto add_tiles = [&](GemmProb t0) {
   for (int c = 0; c < chunks; ++c) {
    GemmProb t = t0;
    if (chunks > 1) {
      const size_t k0 = (size_t)c * 256;
      t.P = t0.P + k0 * t0.ldp; t.Q = t0.Q + k0 * t0.ldq; t.K = 256;
      t.C0 = h->dw_parts + (size_t)c * h->dw_part_stride + (t0.C0 - h->grads);
    }
    for (int m0 = 0; m0 < t.M; m0 += TM)
      for (int n0 = 0; n0 < t.N; n0 += TN) {
        GemmProb q = t;
        q.tiles_n = m0; q.tile_end = n0;  // table form: tile origin
        tiles.push_back(q);
      }
   }
  };
  int which = 0;
  for (int ch : {C_Q1C, C_Q2C, C_PI}) {
    h->dw_off[which++] = (int)tiles.size();
    if (ch == C_Q2C && !twin) continue;   // DSAC_V1: no second critic (empty tile range)
    const int net = kChainNet[ch];
    const NetDesc& d = net_desc(h, net);
    float* g = net_grads(h, net);
    const int slot = kDzSlot[ch];
    const int L = chain_L(h, ch);   // (shadows build_tasks' L: this net's own depth)
    for (int li = 0; li <= L; ++li) {
      // policy: output layer first (see dw_pol_rest); critics: layer order
      const int l = ch == C_PI ? (li == 0 ? L : li - 1) : li;
      if (ch == C_PI && li == 1) h->dw_pol_rest = (int)tiles.size();
      const float* dz; int lddz;
      if (l < L) { dz = h->dZ[slot][l]; lddz = d.out[l]; }
      else if (ch == C_PI) { dz = h->dout_pi; lddz = 2 * h->A; }
      else { dz = h->dout[ch == C_Q1C ? 0 : 1]; lddz = 2; }
      const int nb = (d.nblk == 2 && l > 0) ? 2 : 1;
      const int hb = d.out[l] / nb, kb = d.in[l] / nb;
      for (int b = 0; b < nb; ++b) {
        GemmProb t;
        memset(&t, 0, sizeof(t));
        t.P = dz + (size_t)b * hb; t.ldp = lddz;
        t.M = hb; t.K = B;
        t.Q = l == 0 ? chain_input(h, ch) : h->Hb[ch][l - 1] + (size_t)b * kb;
        t.ldq = l == 0 ? h->ldx : d.in[l];
        t.N = kb;
        if (l < L || nb == 1) { t.C0 = g + d.w_off[l] + (size_t)b * hb * kb; t.ldc = kb; }
        else { t.C0 = g + d.w_off[l] + (size_t)b * hb * d.in[l] + (size_t)b * kb; t.ldc = d.in[l]; }  // [[w_mean,0],[0,w_ls]]
        if (h->d_mir) t.mir = h->d_mir + (size_t)(which - 1) * (L + 1) + l;   // chain mode: packed copies of this tensor
        if (ch == C_PI && l == L && h->cfg.policy_std_param) t.mzero = h->A;   // (see DwProb::msplit for the chain path's tiles)
        add_tiles(t);
      }
      // bias: Q = ones
      GemmProb t;
      memset(&t, 0, sizeof(t));
      t.P = dz; t.ldp = lddz; t.M = d.out[l]; t.K = B;
      t.Q = h->ones; t.ldq = 1; t.N = 1;
      t.C0 = g + d.b_off[l]; t.ldc = 1;
      add_tiles(t);
    }
  }
  if (h->d_tiles) { hipFree(h->d_tiles); h->d_tiles = nullptr; }
  h->n_dw_tiles = (int)tiles.size();
  h->dw_off[3] = h->n_dw_tiles;
  HIPCHK(h, hipMalloc(&h->d_tiles, tiles.size() * sizeof(GemmProb)));
  HIPCHK(h, hipMemcpy(h->d_tiles, tiles.data(), tiles.size() * sizeof(GemmProb), hipMemcpyHostToDevice));
  return DSACT_OK;
}

long long* tl_for(dsact_handle* h, const char* name) {
  return (!h->env_timeline_stage.empty() && h->env_timeline_stage == name) ? h->timeline : nullptr;
}

FusedOpt fused_opt(const dsact_handle* h, bool enable) {
  FusedOpt f;
  memset(&f, 0, sizeof(f));
  f.st = (enable && h->dw_chunks == 1) ? h->st : nullptr;   // split-K partials: the optimiser runs after k_sum_parts
  f.online = h->online; f.target = h->target; f.adam_m = h->adam_m; f.adam_v = h->adam_v; f.grads = h->grads;
  f.n_q2 = (long long)(h->nq * h->n_q); f.n_online3 = (long long)(h->nq * h->n_q + h->n_pi); f.n_total = (long long)h->n_online;
  f.b1w = (float)(1.0 - h->cfg.adam_beta1);
  f.beta2 = (float)h->cfg.adam_beta2;
  f.b2w = (float)(1.0 - h->cfg.adam_beta2);
  f.eps = h->cfg.adam_eps;
  const double polyak = 1.0 - h->cfg.tau;
  f.polyak = (float)polyak;
  f.one_minus_polyak = (float)(1.0 - polyak);
  f.auto_alpha = h->cfg.auto_alpha;
  if (h->mirror_w0 && f.st && !h->chain_ok) {
    // merged-gather graph replays: no per-step repack -- the first-layer tiles of the Q nets keep the copies fresh
    const int on[2] = {N_Q1, N_Q2};
    f.mir_n = h->nq;
    f.mir_ldp = h->ldx; f.mir_O = h->F; f.mir_A = h->A; f.mir_rows = h->wq[0];
    for (int i = 0; i < h->nq; ++i) {
      f.mir_lo[i] = (long long)(net_grads(h, on[i]) - h->grads) + (long long)h->qd.w_off[0];
      f.mir_w[i] = h->W1p[i]; f.mir_wt[i] = h->W1p[2 + i];
      f.mir_at[i] = h->alt.W1aT[i];   // the NEXT update's copy (this one is being read by k_heads_bwd)
    }
  }
  return f;
}

// runs a stage; tiles [x0, x1) of the weight-gradient table ride along in the same launch
int run_stage(dsact_handle* h, const Stage& s0, int x0 = 0, int x1 = 0, bool fused = false) {
  if (s0.n_blocks == 0 && x1 <= x0) return DSACT_OK;
  Stage s = s0;
  s.args.timeline = (!h->env_timeline_stage.empty() && s.name == h->env_timeline_stage) ? h->timeline : nullptr;
  s.args.n_stage_blocks = s.n_blocks;
  s.args.extra = h->d_tiles + x0;
  s.args.n_extra = x1 > x0 ? x1 - x0 : 0;
  s.args.fo = fused_opt(h, fused);
  // large batches: 64x64 tiles (k_stage64) when every problem of the stage allows it and nothing rides along
  // (kind 2, plain store: the conv data gradient's dCol products -- many 32 x 32 tiles with a contraction of only 64-256)
  if (s.args.n_extra == 0 && (s.kind == 0 || s.kind == 1 || (s.kind >= 2 && s.name.compare(0, 9, "conv_dcol") == 0))) {   // (dfeat on 48 such tiles: 14.3 vs 12.3 us)
    bool ok = true;
    int blocks = 0;
    StageArgs a64 = s.args;
    for (int q = 0; q < a64.n_prob && ok; ++q) {
      GemmProb& g = a64.p[q];
      int K = g.K;
      if (K % 4) {   // first layer of the Q nets: both operands are zero padded to ld (k_gather / repack)
        const int K4 = (K + 3) & ~3;
        if (s.kind == 0 && g.ldp >= K4 && g.ldq >= K4) K = K4; else ok = false;
      }
      ok = ok && g.M >= (s.kind >= 2 ? h->env_dcol64_min_m : 512) && g.M % 64 == 0 && g.N % 64 == 0;
      g.K = K;
      g.tiles_n = g.N / 64;
      blocks += (g.M / 64) * g.tiles_n;
      g.tile_end = blocks;
    }
    if (ok) {
      if (s.kind == 0) return launch(h, s.name.c_str(), k_stage64<false, EPI_GELU>, dim3(blocks), dim3(kThreads64), tile64_lds_bytes(), a64);
      if (s.kind == 2) return launch(h, s.name.c_str(), k_stage64<true, EPI_STORE>, dim3(blocks), dim3(kThreads64), tile64_lds_bytes(), a64);
      return launch(h, s.name.c_str(), k_stage64<true, EPI_MULG>, dim3(blocks), dim3(kThreads64), tile64_lds_bytes(), a64);
    }
  }
  const int grid = s.n_blocks + s.args.n_extra;
  size_t lds = tile_lds_bytes(s.max_k);
  if (s.args.n_extra && tile_lds_bytes(dw_k(h)) > lds) lds = tile_lds_bytes(dw_k(h));
  // clean stages (hidden layers of 128 / 256 units at batch sizes that are multiples of 32) use the
  // straight-line specialisations
#define STAGE_TS(PM, QM, EP)                                                                                     \
  do {                                                                                                           \
    if (s.ts == 4) return launch(h, s.name.c_str(), k_stage<PM, QM, EP, 4>, dim3(grid), dim3(kThreads), lds, s.args); \
    if (s.ts == 2) return launch(h, s.name.c_str(), k_stage<PM, QM, EP, 2>, dim3(grid), dim3(kThreads), lds, s.args); \
  } while (0)
  if (s.kind == 0) STAGE_TS(false, false, EPI_GELU);
  if (s.kind == 1) STAGE_TS(false, true, EPI_MULG);
#undef STAGE_TS
  if (s.kind == 0)
    return launch(h, s.name.c_str(), k_stage<false, false, EPI_GELU>, dim3(grid), dim3(kThreads), lds, s.args);
  if (s.kind == 2)
    return launch(h, s.name.c_str(), k_stage<false, true, EPI_STORE>, dim3(grid), dim3(kThreads), lds, s.args);
  return launch(h, s.name.c_str(), k_stage<false, true, EPI_MULG>, dim3(grid), dim3(kThreads), lds, s.args);   // kinds 1 and 3
}

// weight-gradient tiles [x0, x1) as their own launch; `fused`: Adam/Polyak in the tile epilogue;
// `finalize`: one extra block closes the update (alpha step, EMA commit, counters)
int run_dw(dsact_handle* h, int x0, int x1, bool fused, bool finalize, hipStream_t on = nullptr) {
  if (x1 <= x0 && !finalize) return DSACT_OK;
  TableArgs a;
  a.tiles = h->d_tiles + x0; a.n_tiles = x1 > x0 ? x1 - x0 : 0;
  a.fo = fused_opt(h, fused);
  a.finalize = finalize ? 1 : 0;
  a.timeline = tl_for(h, "dW");
  if (on)
    return launch_on(h, on, "dW", k_stage_table, dim3(a.n_tiles + (finalize ? 1 : 0)), dim3(kThreads), tile_lds_bytes(dw_k(h)), a);
  return launch(h, "dW", k_stage_table, dim3(a.n_tiles + (finalize ? 1 : 0)), dim3(kThreads), tile_lds_bytes(dw_k(h)), a);
}


// ---- CNN encoders -----------------------------------------------------------------------------------
// Conv stacks are numbered [q x nq, policy | q_target x nq, policy_target]: nd = nq + 1 differentiated stacks on `obs`, then the
// nd target stacks on `obs2` (DSAC_V2: q1 q2 pi q1t q2t pit = the ConvStack enum; DSAC_V1 with one critic: q pi qt pit)
int n_diff_stacks(const dsact_handle* h) { return h->nq + 1; }
int n_conv_stacks(const dsact_handle* h) { return 2 * (h->nq + 1); }
int stack_net(const dsact_handle* h, int st) {
  const int nq = h->nq, nd = nq + 1;
  if (st < nq) return N_Q1 + st;
  if (st == nq) return N_POL;
  if (st < nd + nq) return N_Q1T + (st - nd);
  return N_POLT;
}
int stack_chain(const dsact_handle* h, int st) {   // the chain whose MLP rows a stack's features feed first
  const int nq = h->nq, nd = nq + 1;
  if (st < nq) return C_Q1C + st;
  if (st == nq) return C_PI;
  if (st < nd + nq) return C_Q1T + (st - nd);
  return C_PIT;
}

// conv stacks forward: one launch per layer carrying all six stacks, then the feature scatter into the
// MLP input rows of the chains each stack feeds
ConvIndex conv_index(const ConvGeom& g) {
  ConvIndex ix;
  ix.OHW = g.OH * g.OW;
  ix.inv_ohw = 1.0f / (float)ix.OHW;
  ix.inv_ow = 1.0f / (float)g.OW;
  return ix;
}

// persistent grid of a forward conv launch: enough workgroups to fill the chip (4 per CU at 36.8 KB LDS)
int conv_fwd_grid(int n_items) { return n_items < 1024 ? n_items : 1024; }

int enqueue_conv_forward(dsact_handle* h) {
  const int B = h->B;
  bool feat_direct = false;   // the last layer's kernel wrote the trunks' feature rows itself (one-pixel stacks on k_conv_fwd64)
  // (round 6) a stack that ends in ONE pixel: features == last-layer activations, in the same order
  auto feat_rows = [&](ConvStageArgs& a, int j) {
    if (j != h->n_conv - 1 || h->cP != 1 || h->env_no_dcol_ident) return;
    for (int q = 0; q < a.n_prob; ++q) {     // (one group per stack at every layer but the first)
      a.p[q].feat[0] = h->Xc[stack_chain(h, q)];
      a.p[q].feat[1] = q < h->nq ? h->Xc[C_Q1P + q] : nullptr;   // q(obs, new_act) shares q(obs, act)'s features
      a.p[q].ldf = h->ldx;
    }
    feat_direct = true;
  };
  for (int j = 0; j < h->n_conv; ++j) {
    const ConvGeom& g = h->cg[j];
    ConvStageArgs a;
    memset(&a, 0, sizeof(a));
    a.g = g; a.ix = conv_index(g);
    const int M = B * g.OH * g.OW;
    if ((long long)M >= (1 << 24)) return fail(h, DSACT_E_INVALID, "batch x output pixels of conv layer %d exceeds 2^24", j);
    int items = 0;
    // layer 0: the three nets on `obs` (resp. `obs2`) read the same image -> one group each, weights
    // concatenated along the channel dimension; deeper layers: one group per stack
    const int nd = n_diff_stacks(h), n_stacks = n_conv_stacks(h);
    const int per_group = j == 0 ? nd : 1;
    for (int st0 = 0; st0 < n_stacks; st0 += per_group) {
      ConvGroup& p = a.p[a.n_prob++];
      p.in = j == 0 ? h->img[st0 < nd ? 0 : 1] : h->cact[st0][j - 1];
      p.n_sub = per_group;
      for (int u = 0; u < per_group; ++u) {
        const int st = st0 + u, net = stack_net(h, st);
        const NetDesc& d = net_desc(h, net);
        p.w[u] = net_params(h, net) + d.cw_off[j];
        p.bias[u] = net_params(h, net) + d.cb_off[j];
        p.out[u] = h->cact[st][j];
      }
      p.M = M;
      p.tiles_n = tiles_of(per_group * g.Cout, TN);
      items += tiles_of(M, TM) * p.tiles_n;
      p.item_end = items;
    }
    a.n_items = items;
    const std::string name = "conv_fwd_l" + std::to_string(j);
    // (K <= 144 with 32 channels -- type_2's third layer: 9 k-groups, 256 VGPRs, ONE wave per SIMD that prefetches the next
    //  tile's 18 patch quads under its 144 MFMAs; round 4: 40.0 us on the LDS-tile kernel)
    const bool narrow9 = g.K > 80 && g.K <= 144 && per_group * g.Cout <= 32 && !h->env_no_conv_narrow9;
    if ((g.K <= 80 || narrow9) && per_group * g.Cout <= 32) {
      // narrow layer: wave-autonomous register tiles (k_conv_fwd_narrow); every wave works on one group
      const int waves_total = narrow9 ? 4 * h->n_cu : 5120;   // ~20 waves per CU; the 9-group form: every SIMD one wave, one round
      int wpg = waves_total / a.n_prob;
      if (wpg > tiles_of(M, 32)) wpg = tiles_of(M, 32);
      a.n_items = wpg;                                    // waves per group
      const int grid = (a.n_prob * wpg + 3) / 4;
      const bool one_block = per_group * g.Cout <= 16;
      if (narrow9) TRY(launch(h, name.c_str(), (k_conv_fwd_narrow<9, 2>), dim3(grid), dim3(kThreads), 0, a));
      else if (g.K <= 48 && one_block) TRY(launch(h, name.c_str(), (k_conv_fwd_narrow<3, 1>), dim3(grid), dim3(kThreads), 0, a));
      else if (g.K <= 48) TRY(launch(h, name.c_str(), (k_conv_fwd_narrow<3, 2>), dim3(grid), dim3(kThreads), 0, a));
      else if (one_block) TRY(launch(h, name.c_str(), (k_conv_fwd_narrow<5, 1>), dim3(grid), dim3(kThreads), 0, a));
      else TRY(launch(h, name.c_str(), (k_conv_fwd_narrow<5, 2>), dim3(grid), dim3(kThreads), 0, a));
      continue;
    }
    // wide layers: 64 x 64 tiles when they divide the problem and still give every CU a workgroup (type_2 layers 3, 4)
    if (per_group == 1 && M % 64 == 0 && g.Cout % 64 == 0 && g.K % 4 == 0 && !h->env_no_conv_fwd64 &&
        (long long)a.n_prob * (M / 64) * (g.Cout / 64) >= h->env_conv_fwd64_min) {
      int it64 = 0;
      for (int q = 0; q < a.n_prob; ++q) { it64 += (M / 64) * (g.Cout / 64); a.p[q].item_end = it64; a.p[q].tiles_n = g.Cout / 64; }
      a.n_items = it64;
      feat_rows(a, j);
      TRY(launch(h, name.c_str(), k_conv_fwd64<2>, dim3(it64), dim3(kThreads64), tile64_lds_bytes(), a));
      continue;
    }
    // ... 32 x 64 tiles where those would leave CUs idle but the contraction is long (type_2 layer 5: K = 1152, M = batch)
    if (per_group == 1 && M % 32 == 0 && g.Cout % 64 == 0 && g.K % 4 == 0 && g.K >= 512 && !h->env_no_conv_fwd64 && !h->env_no_conv_fwd32x64) {
      int it = 0;
      for (int q = 0; q < a.n_prob; ++q) { it += (M / 32) * (g.Cout / 64); a.p[q].item_end = it; a.p[q].tiles_n = g.Cout / 64; }
      a.n_items = it;
      feat_rows(a, j);
      TRY(launch(h, name.c_str(), k_conv_fwd64<1>, dim3(it), dim3(kThreads64), tile64_lds_bytes(), a));
      continue;
    }
    TRY(launch(h, name.c_str(), k_conv_fwd, dim3(conv_fwd_grid(items)), dim3(kThreads), 0, a));
  }
  if (feat_direct) return DSACT_OK;
  FeatArgs f;
  memset(&f, 0, sizeof(f));
  const int last = h->n_conv - 1;
  const int n_stacks = n_conv_stacks(h);
  for (int st = 0; st < n_stacks; ++st) { f.act[st] = h->cact[st][last]; f.dst0[st] = h->Xc[stack_chain(h, st)]; f.dst1[st] = nullptr; }
  for (int i = 0; i < h->nq; ++i) f.dst1[i] = h->Xc[C_Q1P + i];   // q(obs, new_act) shares q(obs, act)'s features
  f.n_stack = n_stacks; f.B = B; f.P = h->cP; f.C = h->cg[last].Cout; f.ldx = h->ldx;
  const long long n = (long long)B * h->F;
  return launch(h, "feat_scatter", k_feat_scatter, dim3((unsigned)((n + kThreads - 1) / kThreads), n_stacks), dim3(kThreads), 0, f);
}

// conv stacks backward for the first n_st differentiated stacks (q1, q2[, policy]); dfeat[] must hold
// dL/d(features). Per layer: weight/bias gradient partials + ordered reduce (+ Adam/Polyak when `fused`),
// then dCol = dY W and the col2im gather into the previous layer's dY.
// stacks [st_lo, st_lo + n_st) of (q1, q2, policy)
// Chunk actually used by layer j's k_conv_dw launch. A k_conv_dw workgroup holds 2 x (1 + NKT) LDS tiles, four fit a CU:
// the chip runs 1,024 of them at a time and a launch of 1,106 (layer 0 at the minimum chunk) takes two full rounds, the
// second one for 82 workgroups (found round 3: 44 us where ~25 are due; neither deeper prefetch nor half the address
// arithmetic nor fewer L1 requests had moved it). Longer chunks mean fewer, longer workgroups: pick the multiple of 64
// that minimises rounds x (start-up + steps) under the measured ~3 us + ~1.1 us per 64-pixel step.
// narrow layers (all channels of a problem in <= 32 rows, K + 4 <= 160 columns): the register-tile weight gradient (k_conv_dw_reg)
bool conv_dw_reg_ok(const dsact_handle* h, int j, int n_st) {
  const ConvGeom& g = h->cg[j];
  const int per_prob = j == 0 ? n_st : 1;
  return ((h->env_conv_dw_reg >> j) & 1) && per_prob * g.Cout <= 32 && g.K + 4 <= 160 && g.OW >= 4 && !h->conv_fork;
}

int conv_dw_pick_chunk(const dsact_handle* h, int j, int n_st) {
  const ConvGeom& g = h->cg[j];
  const long long M = (long long)h->B * g.OH * g.OW;
  const int c0 = (int)conv_dw_chunk((size_t)M);
  if (conv_dw_reg_ok(h, j, n_st)) {
    // one workgroup (4 waves) per chunk and problem: about two workgroups per CU, never more chunks than the partial buffers hold
    const int n_prob = j == 0 ? 1 : n_st;
    const long long want = ((long long)h->env_conv_dw_reg_wgs + n_prob - 1) / n_prob;
    long long c = (M + want - 1) / want;
    c = (c + 63) / 64 * 64;
    return (int)(c < c0 ? c0 : c);
  }
  const int nkt = h->conv_dw_nkt_l[j];
  const int per_prob = j == 0 ? n_st : 1, n_prob = n_st / per_prob;
  const long long per_chunk = (long long)tiles_of(per_prob * g.Cout, TM) * tiles_of(tiles_of(g.K + 4, TN), nkt) * n_prob;
  const long long slots = ((nkt == 1 || (nkt == 3 && h->env_conv_dw_sb3)) ? 4LL : 2LL) * h->n_cu;   // (1 or 2) x (1 + nkt) LDS tiles of 9 KB per workgroup
  int best = c0;
  double best_cost = 1e30;
  for (int c = c0; c <= 16 * c0 && c <= 8192; c += 64) {
    const long long blocks = ((M + c - 1) / c) * per_chunk;
    const long long rounds = (blocks + slots - 1) / slots;
    const double cost = (double)rounds * (3.0 + 1.1 * (c / 64) * (nkt == 1 ? 1.0 : 0.6 + 0.4 * nkt));
    if (cost < best_cost - 1e-9) { best_cost = cost; best = c; }
  }
  return best;
}

int enqueue_conv_backward(dsact_handle* h, int n_st, bool fused, int st_lo = 0) {
  const int B = h->B;
  const int last = h->n_conv - 1;
  auto S = [st_lo](int st) { return st_lo + st; };
  if (!(h->cP == 1 && !h->env_no_dcol_ident)) {   // (one-pixel stacks: the dfeat product wrote the masked dY itself, build_tasks)
    FeatBwdArgs f;
    memset(&f, 0, sizeof(f));
    for (int st = 0; st < n_st; ++st) { f.dfeat[st] = h->dfeat[S(st)]; f.act[st] = h->cact[S(st)][last]; f.dy[st] = h->cdy[S(st)][last]; }
    f.n_stack = n_st; f.B = B; f.P = h->cP; f.C = h->cg[last].Cout;
    const long long n = (long long)B * h->F;
    TRY(launch(h, "feat_bwd", k_feat_bwd, dim3((unsigned)((n + kThreads - 1) / kThreads), n_st), dim3(kThreads), 0, f));
  }
  const bool fork = h->conv_fork && !h->profiling;   // (the per-kernel profile runs everything on one stream)
  if (fork) HIPCHK(h, hipEventRecord(h->ev_conv[last], h->stream));
  for (int j = last; j >= 0; --j) {
    const ConvGeom& g = h->cg[j];
    const int M = B * g.OH * g.OW;
    const std::string sfx = "_l" + std::to_string(j);
    const int chunk = conv_dw_pick_chunk(h, j, n_st), n_chunks = (M + chunk - 1) / chunk, K1p = g.K + 4;
    {
      ConvDwArgs a;
      memset(&a, 0, sizeof(a));
      a.g = g; a.ix = conv_index(g);
      a.chunk = chunk; a.n_chunks = n_chunks; a.K1p = K1p;
      const int kt_all = tiles_of(a.K1p, TN);
      // k-tiles per workgroup (the dY tile would be staged once for all of them): measured slower than one
      // k-tile per workgroup on every layer (fewer, longer dependent chains) -> 1
      const int nkt = h->conv_dw_nkt_l[j];
      a.tiles_k = tiles_of(kt_all, nkt);             // k-groups
      int blocks = 0;
      // layer 0: every differentiated stack reads the staged `obs` image -> one problem, dY rows concatenated
      const int per_prob = j == 0 ? n_st : 1;
      for (int st0 = 0; st0 < n_st; st0 += per_prob) {
        ConvDwProb& p = a.p[a.n_prob++];
        p.in = j == 0 ? h->img[0] : h->cact[S(st0)][j - 1];
        p.n_sub = per_prob;
        for (int u = 0; u < per_prob; ++u) { p.dy[u] = h->cdy[S(st0 + u)][j]; p.part[u] = h->dwpart[S(st0 + u)][j]; }
        p.M = M;
        p.tiles_co = tiles_of(per_prob * g.Cout, TM);
        blocks += a.n_chunks * p.tiles_co * a.tiles_k;
        p.block_end = blocks;
      }
      const bool sb3 = nkt == 3 && h->env_conv_dw_sb3;   // single-buffered three-k-tile form
      const size_t lds = (size_t)(sb3 ? 1 : 2) * (1 + nkt) * TILE_LDS * sizeof(float);
      if (conv_dw_reg_ok(h, j, n_st)) {
        // register tiles: one workgroup per (problem, chunk)
        int rb = 0;
        for (int q = 0; q < a.n_prob; ++q) { rb += a.n_chunks; a.p[q].block_end = rb; }
        const int ncb = (per_prob * g.Cout + 15) / 16, nkb = (a.K1p + 15) / 16;
        const std::string nm = "conv_dw" + sfx;
        if (ncb == 1 && nkb <= 4) TRY(launch(h, nm.c_str(), (k_conv_dw_reg<1, 4>), dim3(rb), dim3(kThreads), 0, a));
        else if (ncb == 2 && nkb <= 4) TRY(launch(h, nm.c_str(), (k_conv_dw_reg<2, 4>), dim3(rb), dim3(kThreads), 0, a));
        else if (ncb == 1 && nkb <= 5) TRY(launch(h, nm.c_str(), (k_conv_dw_reg<1, 5>), dim3(rb), dim3(kThreads), 0, a));
        else if (ncb == 2 && nkb <= 5) TRY(launch(h, nm.c_str(), (k_conv_dw_reg<2, 5>), dim3(rb), dim3(kThreads), 0, a));
        else if (ncb == 1) TRY(launch(h, nm.c_str(), (k_conv_dw_reg<1, 10>), dim3(rb), dim3(kThreads), 0, a));
        else TRY(launch(h, nm.c_str(), (k_conv_dw_reg<2, 10>), dim3(rb), dim3(kThreads), 0, a));
      } else
      if (fork) {
        HIPCHK(h, hipStreamWaitEvent(h->aux_stream, h->ev_conv[j], 0));
        if (nkt != 1) return fail(h, DSACT_E_INVALID, "DSACT_CONV_FORK needs DSACT_CONV_DW_NKT=1");
        TRY(launch_on(h, h->aux_stream, ("conv_dw" + sfx).c_str(), k_conv_dw<1>, dim3(blocks), dim3(kThreads), lds, a));
      } else
      // (three / four steps of loads in flight, k_conv_dw<1, 3|4>: 46.4-48.3 us vs 45 us on layers 0 / 1 -- not latency-bound)
      if (nkt == 1) TRY(launch(h, ("conv_dw" + sfx).c_str(), k_conv_dw<1>, dim3(blocks), dim3(kThreads), lds, a));
      else if (nkt == 2) TRY(launch(h, ("conv_dw" + sfx).c_str(), k_conv_dw<2>, dim3(blocks), dim3(kThreads), lds, a));
      else if (sb3) TRY(launch(h, ("conv_dw" + sfx).c_str(), (k_conv_dw<3, 2, false>), dim3(blocks), dim3(kThreads), lds, a));
      else if (nkt == 3) TRY(launch(h, ("conv_dw" + sfx).c_str(), k_conv_dw<3>, dim3(blocks), dim3(kThreads), lds, a));
      else return fail(h, DSACT_E_INVALID, "DSACT_CONV_DW_NKT must be 1..3");
    }
    // (the 16-channel layer through dCol + col2im instead: 49.5 + 23.8 us vs 42 us direct, measured round 3)
    const bool direct_dx = j > 0 && (g.Cin == 8 || g.Cin == 16) && g.Cout <= 32;
    if (direct_dx) {
      // narrow layers: no column buffer (see k_conv_dx_direct)
      ConvDxArgs c;
      memset(&c, 0, sizeof(c));
      c.g = g; c.n_prob = n_st; c.B = B;
      for (int st = 0; st < n_st; ++st) {
        const int net = stack_net(h, S(st));
        c.dy[st] = h->cdy[S(st)][j]; c.w[st] = net_params(h, net) + net_desc(h, net).cw_off[j];
        c.x[st] = h->cact[S(st)][j - 1]; c.dx[st] = h->cdy[S(st)][j - 1];
      }
      const int Yq = (g.H + g.stride - 1) / g.stride, Xq = (g.W + g.stride - 1) / g.stride;   // largest parity class
      if (g.KS == 3 && g.stride == 2 && g.Cin == 16 && g.Cout == 32 && !h->env_no_conv_dx_mfma) {
        // matrix-core variant (dsact_conv.h: k_conv_dx_mfma): two 16-pixel-pair tiles per wave. 16-channel layer: 43.9 ->
        // 24.3 us; the 8-channel layer (half of every tile's columns idle) 30.9 -> 38.6 us, so it keeps the block kernel
        const int Xq_ = (g.W + 1) / 2, Yc0 = (g.H + 1) / 2;
        const int tiles0 = (B * Yc0 * Xq_ + 15) / 16;
        const dim3 gm((unsigned)((tiles0 + 7) / 8), n_st, 2);
        TRY(launch(h, ("conv_dx" + sfx).c_str(), (k_conv_dx_mfma<16, 32>), gm, dim3(kThreads), 0, c));
      } else if (g.KS == 3 && g.stride == 2) {
        // one thread per 2x2 pixel block (all four parity classes); (one thread per pixel on the 16-channel layer, 4x the
        // workgroups: 72.5 us vs 42.6 us, round 3)
        const dim3 gb((unsigned)((B * Yq * Xq + kThreads - 1) / kThreads), n_st);
        if (g.Cin == 8) TRY(launch(h, ("conv_dx" + sfx).c_str(), (k_conv_dx_block<2, 3, 2>), gb, dim3(kThreads), 0, c));
        else TRY(launch(h, ("conv_dx" + sfx).c_str(), (k_conv_dx_block<4, 3, 2>), gb, dim3(kThreads), 0, c));
      } else {
        const dim3 grid((unsigned)((B * Yq * Xq + kThreads - 1) / kThreads), n_st, g.stride * g.stride);
        if (g.Cin == 8) TRY(launch(h, ("conv_dx" + sfx).c_str(), k_conv_dx_direct<2>, grid, dim3(kThreads), 0, c));
        else TRY(launch(h, ("conv_dx" + sfx).c_str(), k_conv_dx_direct<4>, grid, dim3(kThreads), 0, c));
      }
    } else if (j > 0) {
      // dCol[m][k] = sum_co dY[m][co] W[co][k]: dense KC x MC product, plain store
      // A layer with ONE output pixel whose kernel covers its whole input (type_2's last layer: 3 x 3 x 128 -> 1 x 1 x 256): every
      // input pixel receives exactly one column of dCol, in dCol's own order -- col2im is the identity, and the ReLU mask of the
      // layer's input is applied in the product's epilogue (MULG_RELU_MASK) straight into the previous layer's dY: one launch
      // less, bit-identical values (round 6)
      const bool ident = g.OH == 1 && g.OW == 1 && g.H == g.KS && g.W == g.KS && !h->env_no_dcol_ident;
      Stage s;
      s.name = "conv_dcol" + sfx; s.kind = ident ? 3 : 2; s.n_blocks = 0;
      memset(&s.args, 0, sizeof(s.args));
      for (int st = 0; st < n_st; ++st) {
        const int net = stack_net(h, S(st));
        const NetDesc& d = net_desc(h, net);
        GemmProb t;
        memset(&t, 0, sizeof(t));
        t.P = h->cdy[S(st)][j]; t.ldp = g.Cout;
        t.Q = net_params(h, net) + d.cw_off[j]; t.ldq = g.K;
        t.C0 = h->dcol[S(st)]; t.ldc = g.K;
        t.M = M; t.N = g.K; t.K = g.Cout;
        if (ident) { t.C0 = h->cdy[S(st)][j - 1]; t.aux = h->cact[S(st)][j - 1]; t.ldaux = g.K; t.act = MULG_RELU_MASK; }
        stage_add(s, t);
      }
      TRY(run_stage(h, s));
      if (ident) { if (fork && j > 0) HIPCHK(h, hipEventRecord(h->ev_conv[j - 1], h->stream)); continue; }
      Col2imArgs c;
      memset(&c, 0, sizeof(c));
      c.g = g; c.n_prob = n_st; c.B = B;
      for (int st = 0; st < n_st; ++st) { c.dcol[st] = h->dcol[S(st)]; c.x[st] = h->cact[S(st)][j - 1]; c.dx[st] = h->cdy[S(st)][j - 1]; }
      const long long n = (long long)B * g.H * g.W * (g.Cin / 4);
      TRY(launch(h, ("col2im" + sfx).c_str(), k_col2im, dim3((unsigned)((n + kThreads - 1) / kThreads), n_st), dim3(kThreads), 0, c));
    }
    if (fork && j > 0) HIPCHK(h, hipEventRecord(h->ev_conv[j - 1], h->stream));   // dY[j-1] is complete
  }
  if (fork) {
    HIPCHK(h, hipEventRecord(h->ev_join, h->aux_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  }
  {
    // ordered reduce of the partials of ALL layers (+ Adam / Polyak when fused) in one launch, after the whole conv
    // backward: every dCol / direct data gradient above read the weights as the forward pass saw them
    ConvReduceArgs r;
    memset(&r, 0, sizeof(r));
    int blocks = 0;
    for (int j = 0; j <= last; ++j) {
      const ConvGeom& g = h->cg[j];
      const int M = B * g.OH * g.OW;
      const int chunk = conv_dw_pick_chunk(h, j, n_st);
      ConvReduceLayer& Ly = r.L[j];
      Ly.Cout = g.Cout; Ly.K = g.K; Ly.K1p = g.K + 4; Ly.n_chunks = (M + chunk - 1) / chunk;
      Ly.quads = g.Cout * Ly.K1p / 4;
      Ly.wide = Ly.n_chunks > 16 ? 1 : 0;
      Ly.block_begin = blocks;
      blocks += Ly.wide ? (Ly.quads + 15) / 16 : (Ly.quads + kThreads - 1) / kThreads;
      for (int st = 0; st < n_st; ++st) {
        const int net = stack_net(h, S(st));
        const NetDesc& d = net_desc(h, net);
        r.p[j][st].part = h->dwpart[S(st)][j];
        r.p[j][st].w_idx = (long long)((net_grads(h, net) + d.cw_off[j]) - h->grads);
        r.p[j][st].b_idx = (long long)((net_grads(h, net) + d.cb_off[j]) - h->grads);
      }
    }
    r.n_layers = last + 1; r.n_prob = n_st;
    r.fo = fused_opt(h, fused);
    TRY(launch(h, "conv_dw_reduce", k_conv_dw_reduce, dim3(blocks, n_st), dim3(kThreads), 0, r));
  }
  return DSACT_OK;
}

// image rows -> staged pixel-major minibatch (+ replayed action / reward / done when `with_scalars`)
RepackArgs repack_args(const dsact_handle* h, int n_blocks);
int repack_blocks(const dsact_handle* h);
StepHyper step_hyper(const dsact_handle* h);
NoiseArgs noise_args(const dsact_handle* h);

// step_it >= -1: fused step flow (bookkeeping for iteration step_it, or the device iteration when use_dev; device
// noise; weight repack) rides in the same launch; -2: plain staging
int enqueue_gather_img(dsact_handle* h, const float* src_obs, const float* src_obs2, const int* table, int rows,
                       int use_dev, bool with_scalars, float* img0, float* img2, int n_rows,
                       long long step_it = -2, int advance = 0) {
  ImgGatherArgs a;
  memset(&a, 0, sizeof(a));
  if (step_it >= -1) {
    a.bookkeeping = 1; a.advance_counters = advance; a.host_it = step_it; a.stw = h->st;
    a.hp = step_hyper(h); a.nz = noise_args(h); a.rp = repack_args(h, repack_blocks(h));
  }
  a.rb_obs = src_obs; a.rb_obs2 = src_obs2;
  a.rb_act = with_scalars ? h->rb_act : nullptr; a.rb_rew = h->rb_rew; a.rb_done = h->rb_done;
  a.idx_table = table; a.idx_rows = rows; a.use_dev = use_dev; a.host_row = 0; a.st = h->st;
  a.img0 = img0; a.img2 = img2; a.Xa0 = h->Xc[C_Q1C]; a.Xa1 = h->Xc[C_Q2C]; a.rew = h->rew; a.done = h->done;
  a.B = n_rows; a.C = h->cfg.img_c; a.HW = h->cfg.img_h * h->cfg.img_w; a.A = h->A; a.F = h->F; a.ldx = h->ldx;
  a.chunks = 8;
  if (a.C == 3 && a.HW % 4 == 0) {   // RGB fast path: a thread per pixel quad -- exactly one trip per thread when it divides (96 x 96: 9 blocks)
    const int per = (a.HW / 4 + kThreads - 1) / kThreads;
    a.chunks = per < 1 ? 1 : (per > 16 ? 16 : per);
  }
  return launch(h, "gather_img", k_gather_img, dim3(n_rows * a.chunks + a.rp.n_blocks), dim3(kThreads), 0, a);
}

// dispatch on the number of 256-wide chunks of a hidden row (register arrays are statically indexed)
#define NCH_DISPATCH(W, CALL)                  \
  do {                                         \
    const int nch_ = ((W) + 255) / 256;        \
    if (nch_ <= 1) { CALL(1); }                \
    else if (nch_ == 2) { CALL(2); }           \
    else if (nch_ == 3) { CALL(3); }           \
    else { CALL(4); }                          \
  } while (0)

StepHyper step_hyper(const dsact_handle* h) {
  StepHyper hp;
  hp.delay_update = h->cfg.delay_update;
  hp.lr_q = h->cfg.lr_q; hp.lr_pi = h->cfg.lr_pi; hp.lr_alpha = h->cfg.lr_alpha;
  hp.beta1 = h->cfg.adam_beta1; hp.beta2 = h->cfg.adam_beta2;
  return hp;
}
NoiseArgs noise_args(const dsact_handle* h) {
  NoiseArgs nz;
  nz.seed = h->noise_table_on ? 1 : h->rng_seed;
  nz.eps_new = h->eps_new; nz.eps_2 = h->eps_2; nz.z5 = h->z5; nz.z6 = h->z6;
  nz.table = h->noise_table_on ? h->noise_table : nullptr; nz.B = h->B;
  return nz;
}

RepackArgs repack_args(const dsact_handle* h, int n_blocks) {
  RepackArgs rp;
  const int nets[4] = {N_Q1, N_Q2, N_Q1T, N_Q2T};
  for (int i = 0; i < 4; ++i) { rp.src[i] = net_params(h, nets[i]) + h->qd.w_off[0]; rp.dst[i] = h->W1p[i]; }
  rp.rows = h->wq[0]; rp.K = h->F + h->A; rp.ldp = h->ldx;
  rp.n_blocks = n_blocks;
  rp.w1at[0] = h->W1aT[0]; rp.w1at[1] = h->W1aT[1]; rp.O = h->F; rp.A = h->A;
  rp.skip_pad = h->use_w1p ? 0 : 1;
  rp.pk_jobs = nullptr; rp.pk_n_jobs = 0;
  if (h->chain_ok && n_blocks > 0) { rp.pk_jobs = h->d_pack; rp.pk_n_jobs = h->n_pack_jobs; rp.n_blocks = h->pack_blocks; }
  return rp;
}
int repack_blocks(const dsact_handle* h) {
  if (h->chain_ok) return h->pack_blocks;   // one block per 16 source rows of every weight tensor
  const int total4 = h->use_w1p ? 4 * h->wq[0] * h->ldx : 2 * 32 * h->wq[0];
  int nb = (total4 + kThreads * 8 - 1) / (kThreads * 8);
  // the repack blocks share the gather launch: keep the launch within one round of workgroups (256 CUs)
  const int gather_blocks = h->cnn ? 0 : (h->B + 3) / 4;
  const int cap = gather_blocks < 224 ? 256 - gather_blocks : 32;
  return nb < 1 ? 1 : (nb > cap ? cap : nb);
}

int enqueue_prologue(dsact_handle* h, int use_dev, long long it, int advance, int fill_noise);

// gather arguments of the SELECTED batch set (no repack blocks)
GatherArgs gather_args(const dsact_handle* h, const int* table, int rows, int use_dev, long long it, int advance);

int enqueue_gather(dsact_handle* h, const int* table, int rows, int use_dev, long long it, int advance, int bookkeeping = 1) {
  if (h->cnn) {
    // bookkeeping, device noise and the padded-weight repack ride in the same launch
    return enqueue_gather_img(h, h->rb_obs, h->rb_obs2, table, rows, use_dev, true, h->img[0], h->img[1], h->B,
                              use_dev ? -1 : it, advance);
  }
  GatherArgs a = gather_args(h, table, rows, use_dev, it, advance);
  a.bookkeeping = bookkeeping;
  a.rp = repack_args(h, repack_blocks(h));
  return launch(h, "gather", k_gather, dim3(a.n_gather_blocks + a.rp.n_blocks), dim3(kThreads), 0, a);
}

GatherArgs gather_args(const dsact_handle* h, const int* table, int rows, int use_dev, long long it, int advance) {
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.rb_obs = h->rb_obs; a.rb_obs2 = h->rb_obs2; a.rb_act = h->rb_act; a.rb_rew = h->rb_rew; a.rb_done = h->rb_done;
  a.idx_table = table; a.idx_rows = rows; a.use_dev = use_dev; a.host_it = it; a.host_row = 0;
  a.X0 = h->X0; a.XP = h->XP; a.X2 = h->X2; a.rew = h->rew; a.done = h->done;
  a.B = h->B; a.O = h->O; a.A = h->A; a.ldx = h->ldx;
  a.st = h->st; a.bookkeeping = 1; a.advance_counters = advance; a.hp = step_hyper(h); a.nz = noise_args(h);
  a.n_gather_blocks = (h->B + 3) / 4;
  return a;
}

int enqueue_prologue(dsact_handle* h, int use_dev, long long it, int advance, int fill_noise) {
  PrologueArgs a;
  a.st = h->st; a.use_dev = use_dev; a.host_it = it; a.advance_counters = advance; a.fill_noise = fill_noise;
  a.hp = step_hyper(h); a.nz = noise_args(h); a.B = h->B; a.A = h->A; a.table_rows = h->idx_rows > 0 ? h->idx_rows : 1;
  TRY(launch(h, "prologue", k_prologue, dim3(1), dim3(kThreads), 0, a));
  if ((fill_noise || !advance) && h->chain_ok) return enqueue_pack(h);   // a forward pass follows: refresh the packed copies
  if (fill_noise || !advance) {  // a forward pass follows: refresh the padded first-layer weights
    const int nb = repack_blocks(h);
    if (nb) TRY(launch(h, "repack", k_repack, dim3(nb), dim3(kThreads), 0, repack_args(h, nb)));
  }
  return DSACT_OK;
}

// everything of __compute_gradient after the minibatch is staged (dsac_v2.py:150-206)
int enqueue_adam(dsact_handle* h, bool from_parts = false);

int sum_parts_range(dsact_handle* h, size_t lo, size_t hi);

// split-K: gradient arena [lo, hi) = sum of the chunk partials -- per net, skipping the conv parameters of the CNN
// nets (their gradients are reduced by k_conv_dw_reduce straight into the arena)
int sum_parts(dsact_handle* h, size_t lo, size_t hi) {
  if (h->dw_chunks == 1 || hi <= lo) return DSACT_OK;
  if (!h->cnn) return sum_parts_range(h, lo, hi);
  const int nets[3] = {N_Q1, N_Q2, N_POL};
  for (int i = 0; i < 3; ++i) {
    const NetDesc& d = net_desc(h, nets[i]);
    const size_t b = (size_t)(net_grads(h, nets[i]) - h->grads);
    const size_t s0 = b + d.w_off[0], s1 = b + d.count;
    const size_t x0 = s0 > lo ? s0 : lo, x1 = s1 < hi ? s1 : hi;
    if (x1 > x0) TRY(sum_parts_range(h, x0, x1));
  }
  return DSACT_OK;
}

int sum_parts_range(dsact_handle* h, size_t lo, size_t hi) {
  SumPartsArgs a;
  a.part = h->dw_parts + lo; a.stride = (long long)h->dw_part_stride; a.n_part = h->dw_chunks;
  a.g = h->grads + lo; a.n = (long long)(hi - lo);
  const long long quads = (a.n + 3) / 4;
  return launch(h, "sum_parts", k_sum_parts, dim3((unsigned)((quads + kThreads - 1) / kThreads)), dim3(kThreads), 0, a);
}


// ---- dw2: weight-gradient tiles on the transposed packs the chain kernels write --------------------------------
Dw2Args dw2_args(dsact_handle* h, bool fused) {
  Dw2Args a;
  memset(&a, 0, sizeof(a));
  const int L = h->L;
  const int chs[3] = {C_Q1C, C_Q2C, C_PI};
  int tiles = 0;
  for (int n3 = 0; n3 < 3; ++n3) {
    h->dw2_off[n3] = tiles;
    if (n3 == 1 && h->nq == 1) continue;   // one critic (DSAC_V1): q2's tile range is empty
    const int ch = chs[n3], net = kChainNet[ch], slot = kDzSlot[ch];
    const NetDesc& d = net_desc(h, net);
    const long long base = (long long)(net_grads(h, net) - h->grads);
    if (h->twin) continue;   // (below)
    for (int l = 0; l <= L; ++l) {
      DwProb& P = a.p[a.n_prob++];
      P.At = l < L ? h->dZ[slot][l] : h->doutT[n3];
      P.Xt = l == 0 ? h->X0t : h->Hb[ch][l - 1];
      P.M = d.out[l]; P.N = d.in[l];
      P.w_idx = base + (long long)d.w_off[l]; P.b_idx = base + (long long)d.b_off[l];
      P.tiles_n = (P.N + 31) / 32;
      tiles += ((P.M + 31) / 32) * P.tiles_n;
      P.tile_end = tiles;
      P.mir = h->d_mir ? h->d_mir + (size_t)n3 * (L + 1) + l : nullptr;
      // policy_std_type "parameter": rows [A, 2A) of the policy's output layer are structurally zero -- their gradient is
      // masked (every column: nsplit = N), so Adam / Polyak leave them at 0; the bias rows [A, 2A) are log_std (not masked)
      if (n3 == 2 && l == L && h->cfg.policy_std_param) { P.msplit = h->A; P.nsplit = P.N; }
    }
  }
  if (h->twin) {
    // first layer: dense [2W x in] (both trunks' dZ[0] are one pack of 2W features); hidden layers: one [W x W] problem per
    // trunk; output layer: the dense [n_out x 2W] matrix with its two structurally-zero blocks masked (DwProb::nsplit).
    // Order: critics' hidden + output layers, critics' FIRST layers, policy -- the first-layer tiles (which rewrite W0)
    // can then run behind the ONE launch that forms dL/d features = dZ0 . W0 for all three nets (enqueue_grads_chain)
    const int W = h->cW;
    const size_t tB = (size_t)W * h->B;   // floats between the trunks' halves of an activation pack
    a.n_prob = 0; tiles = 0;
    auto add = [&](int n3, int l, int t) {
      const int ch = chs[n3], net = kChainNet[ch], slot = kDzSlot[ch];
      const NetDesc& d = net_desc(h, net);
      const long long base = (long long)(net_grads(h, net) - h->grads);
      DwProb& P = a.p[a.n_prob++];
      const bool hid = l >= 1 && l < L;
      P.At = l < L ? h->dZ[slot][l] + (hid ? t * tB : 0) : h->doutT[n3];
      P.Xt = l == 0 ? h->X0t_net[n3] : h->Hb[ch][l - 1] + (hid ? t * tB : 0);
      P.M = hid ? W : d.out[l]; P.N = hid ? W : d.in[l];
      P.w_idx = base + (long long)d.w_off[l] + (hid ? (long long)t * W * W : 0);
      P.b_idx = base + (long long)d.b_off[l] + (hid ? (long long)t * W : 0);
      if (l == L) { P.msplit = d.out[L] / 2; P.nsplit = W; }
      P.tiles_n = (P.N + 31) / 32;
      tiles += ((P.M + 31) / 32) * P.tiles_n;
      P.tile_end = tiles;
      P.mir = nullptr;
    };
    for (int n3 = 0; n3 < h->nq; ++n3) {
      h->dw2_off[n3] = tiles;
      for (int l = 1; l <= L; ++l)
        for (int t = 0; t < (l < L ? 2 : 1); ++t) add(n3, l, t);
    }
    if (h->nq == 1) h->dw2_off[1] = tiles;
    h->dw2_mid = tiles;
    for (int n3 = 0; n3 < h->nq; ++n3) add(n3, 0, 0);
    h->dw2_off[2] = tiles;
    for (int l = 0; l <= L; ++l)
      for (int t = 0; t < ((l >= 1 && l < L) ? 2 : 1); ++t) add(2, l, t);
  }
  h->dw2_off[3] = tiles;
  for (int q = 0; q < kMaxDwProb; ++q) a.tile_ends[q] = q < a.n_prob ? a.p[q].tile_end : tiles;
  a.C = h->B / 16;
  a.ct = a.C / h->dw_chunks;   // chunks per batch range (batch <= 256: one round of <= 16; else rounds of 16 inside the tile)
  a.n_base = tiles;
  a.gout = h->dw_chunks > 1 ? h->dw_parts : h->grads;
  a.part_stride = h->dw_chunks > 1 ? (long long)h->dw_part_stride : 0;
  a.fo = fused_opt(h, fused);
  a.store_g = !(fused && h->mirror_w0 && h->dw_chunks == 1);
  return a;
}

// base tiles [x0, x1) of every batch range; `finalize`: one extra block closes the update
int run_dw2(dsact_handle* h, int x0, int x1, bool fused, bool finalize) {
  if (x1 <= x0 && !finalize) return DSACT_OK;
  Dw2Launch L;
  L.a = dw2_args(h, fused);
  L.tile0 = x0; L.n_tiles = x1 > x0 ? x1 - x0 : 0; L.finalize = finalize ? 1 : 0;
  // (k_dw2<4>, fragments four half rounds ahead, measured at batch 512 / 1024: 8.76 vs 8.91 us and 13.10 vs 13.21 us --
  //  the tiles are MFMA-bound there, not load-latency-bound; not instantiated)
  // long contractions (batch >= 512 per range) with one tile per CU: 8 waves per tile, two per SIMD (dsact_chain.h: dw2_tile NWV)
  // (batch 1024: 13.8 -> 11.8 us, 8,611 -> 8,787 steps/s; batch 4096 with its 960 split-K tiles -- several per CU anyway --
  //  is 0.5 % slower with it, hence the tile-count condition; profiles/r03_ab_dw_8wave.txt)
  if (L.a.ct >= 32 && (long long)L.n_tiles * h->dw_chunks <= 512)
    return launch(h, "dW", (k_dw2<2, 8>), dim3(xcd_chunk_grid(L.n_tiles) + (finalize ? 1 : 0), h->dw_chunks), dim3(512), 0, L);
  return launch(h, "dW", k_dw2<2>, dim3(xcd_chunk_grid(L.n_tiles) + (finalize ? 1 : 0), h->dw_chunks), dim3(kThreads), 0, L);
}

// ---- row-slice fused update (dsact_chain.h) ---------------------------------------------------------------------
#define CHAIN_NT(CALL, RGV)                    \
  do {                                         \
    if ((RGV) == 1) {                          \
      if (h->cNT == 1) { CALL(1, 1); }         \
      else if (h->cNT == 2) { CALL(2, 1); }    \
      else { CALL(4, 1); }                     \
    } else if ((RGV) == 4) {                   \
      if (h->cNT == 1) { CALL(1, 4); }         \
      else if (h->cNT == 2) { CALL(2, 4); }    \
      else { CALL(4, 4); }                     \
    } else {                                   \
      if (h->cNT == 1) { CALL(1, 2); }         \
      else if (h->cNT == 2) { CALL(2, 2); }    \
      else { CALL(4, 2); }                     \
    }                                          \
  } while (0)

// Row groups (of 4 rows) per chain workgroup of one launch. A workgroup streams its unit's whole weight set whatever
// its row count, and a step costs ~45 cycles on top of its 32 * RG MFMA cycles: 4-row workgroups finish a layer in
// 0.75x the time of 8-row ones -- as long as every workgroup still gets a CU of its own (n_units * B/4 <= 256 CUs).
// Large batches: 16-row workgroups (RG 4) halve the weight bytes per FLOP. Measured at batch 4096 / 1024: the critics'
// backward gains (81.5 -> 63.7 us / 21.8 -> 20.3 us), the forward launches lose (209 -> 228 us: one fat workgroup per CU
// hides less latency than two 8-row ones), the policy backward is even -- so only the critics' backward asks for it
// (`allow4`), and only when even 16-row slices give every CU a workgroup.
int chain_rg(const dsact_handle* h, int n_units, bool allow4 = false) {
  if (h->env_chain_rg) return h->env_chain_rg;
  if (n_units * (h->B / 4) <= 256) return 1;
  if (allow4 && h->rg4_ok && n_units * (h->B / 16) >= 256) return 4;
  return h->cRG;
}

FwdUnit fwd_unit(const dsact_handle* h, int ch, int seg, int head) {
  FwdUnit u;
  memset(&u, 0, sizeof(u));
  const int net = kChainNet[ch];
  const NetDesc& d = net_desc(h, net);
  const float* base = net_params(h, net);
  for (int l = 0; l <= h->L; ++l) { u.wf[l] = h->pk_fwd[net][l]; u.bias[l] = base + d.b_off[l]; }
  u.x = h->Xc[ch];
  u.seg = seg;
  u.s_act = (net == N_POL || net == N_POLT) ? 0 : (h->fat ? h->c_act : h->s_act);
  if (seg != SEG_OBS_ONLY)
    for (int l = 0; l < h->L; ++l) { u.H[l] = h->Hb[ch][l]; u.G[l] = h->Gb[ch][l]; }
  u.head = head;
  if (head == HEAD_Q && h->cfg.value_out_act) u.head |= h->cfg.value_out_act << HEAD_OUT_ACT_SHIFT;
  if (head == HEAD_POLICY && h->cfg.policy_out_act)
    u.head |= (h->cfg.policy_out_act << HEAD_OUT_ACT_SHIFT) | (h->cfg.policy_std_param ? HEAD_STD_PLAIN : 0);
  u.act = net_act(h, net);
  return u;
}

// rg: rows per workgroup / 4 of every unit that has not chosen its own (FwdUnit::rg != 0); map: nullptr = uniform placement
void fill_fwd_common(dsact_handle* h, FwdArgs& a, int rg, const char* name, const XcdMap* map = nullptr) {
  for (int k = 0; k < a.n_units; ++k) {
    if (!a.u[k].rg) a.u[k].rg = rg;
    a.u[k].n_slices = (short)(h->B / (4 * a.u[k].rg));
  }
  a.map = map ? *map : xcd_map_uniform(a.n_units);
  a.B = h->B; a.F = h->F; a.A = h->A; a.L = h->L; a.ldx = h->ldx;
  a.s_obs = h->fat ? h->c_obs : h->s_obs; a.s_act = h->fat ? h->c_act : h->s_act; a.v1_stats = h->nq == 1; a.Cb = h->B / 16;
  a.act_scale = h->act_scale; a.act_center = h->act_center; a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std;
  a.timeline = tl_for(h, name);
  a.spin_timeout = h->handoff_dev;
  a.debug_withhold = h->debug_withhold == 1;
  a.tpad = h->fat ? 0 : h->env_pk_pad;
}

// group A: policy(obs), policy_target(obs2), q1/q2(obs,act) + observation part of q1_t/q2_t(obs2, .)
void fwd_args_a(dsact_handle* h, FwdArgs& a) {
  memset(&a, 0, sizeof(a));
  FwdUnit& pi = a.u[0] = fwd_unit(h, C_PI, SEG_FULL, HEAD_POLICY);
  pi.logits = h->logits_pi; pi.logp = h->logp_new; pi.eps = h->eps_new; pi.xact = h->Xc[C_Q1P]; pi.part_heads = h->part_heads;
  FwdUnit& pt = a.u[1] = fwd_unit(h, C_PIT, SEG_FULL, HEAD_POLICY);
  pt.logits = h->logits_pit; pt.logp = h->logp2; pt.eps = h->eps_2; pt.xact = h->Xc[C_Q1T];
  for (int l = 0; l < h->L; ++l) pt.G[l] = nullptr;   // never differentiated
  const int nq = h->nq;      // 2 critics (DSAC_V2) or 1 (DSAC_V1): units [pi, pit, q_c x nq, q_t(obs) x nq]
  for (int i = 0; i < nq; ++i) {
    FwdUnit& qc = a.u[2 + i] = fwd_unit(h, C_Q1C + i, SEG_FULL_SAVE, HEAD_Q);
    qc.zsave = h->zobs[i]; qc.qout = h->qout_c[i]; qc.qstd = h->qstd_c[i];
    if (i == 0) qc.x0t = h->X0t;
    FwdUnit& qt = a.u[2 + nq + i] = fwd_unit(h, C_Q1T + i, SEG_OBS_ONLY, HEAD_NONE);
    qt.zsave = h->zobs[2 + i];
  }
  a.n_units = 2 + 2 * nq;
}

// group B: q1_t/q2_t(obs2,act2), q1/q2(obs,new_act): saved observation part + action part, hidden layers, heads
void fwd_args_b(dsact_handle* h, FwdArgs& a) {
  memset(&a, 0, sizeof(a));
  const int nq = h->nq;      // units [q_t x nq, q_p x nq]
  for (int i = 0; i < nq; ++i) {
    FwdUnit& qt = a.u[i] = fwd_unit(h, C_Q1T + i, SEG_ACT_FROM_SAVED, HEAD_Q);
    qt.zinit = h->zobs[2 + i]; qt.qout = h->qout_t[i];
    for (int l = 0; l < h->L; ++l) qt.G[l] = nullptr;   // never differentiated
    FwdUnit& qp = a.u[nq + i] = fwd_unit(h, C_Q1P + i, SEG_ACT_FROM_SAVED, HEAD_Q);
    qp.zinit = h->zobs[i]; qp.qout = h->qout_p[i];
  }
  a.n_units = 2 * nq;
}

// ---- twin-trunk nets (CNN approximators): forward launches from device-memory tables (k_chain_fwdt) -----------------
// unit of trunk t of chain ch: the trunk's half of every twin-width buffer (see dsact_handle::twin)
FwdUnit fwd_unit_twin(const dsact_handle* h, int ch, int t, int head_kind) {
  FwdUnit u;
  memset(&u, 0, sizeof(u));
  const int net = kChainNet[ch], L = h->L, W = h->cW, tiles = W / 64, SH = W / 4, CH = W / 16;
  const bool pol = net == N_POL || net == N_POLT;
  const NetDesc& d = net_desc(h, net);
  const float* base = net_params(h, net);
  const int S0 = h->s_obs + (pol ? 0 : h->s_act);
  u.wf[0] = h->pk_fwd[net][0] + (size_t)t * tiles * S0 * 256;
  for (int l = 1; l < L; ++l) u.wf[l] = h->pk_fwd[net][l] + (size_t)t * tiles * SH * 256;
  u.wf[L] = h->pk_fwd[net][L] + (size_t)t * CH * 256;
  for (int l = 0; l < L; ++l) u.bias[l] = base + d.b_off[l] + (size_t)t * W;
  u.bias[L] = base + d.b_off[L];
  u.x = h->Xc[ch];
  u.seg = SEG_FULL;
  u.s_act = pol ? 0 : h->s_act;
  for (int l = 0; l < L; ++l) { u.H[l] = h->Hb[ch][l] + (size_t)t * W * h->B; u.G[l] = h->Gb[ch][l] + (size_t)t * W * h->B; }
  u.head = head_code(head_kind, t == 0 ? HEAD_TWIN_FIRST : HEAD_TWIN_SECOND, 2 * CH);
  u.act = (short)net_act(h, net);
  return u;
}

// group 0: policy(obs), policy_target(obs2), q_c(obs, act) x nq; group 1: q_t(obs2, act2) x nq, q(obs, new_act) x nq
int build_twin_fwd(dsact_handle* h) {
  const int nq = h->nq, L = h->L;
  for (int grp = 0; grp < 2; ++grp) {
    if (!h->fwdt_host[grp]) h->fwdt_host[grp] = new PipeFwd;
    if (!h->d_fwdt[grp]) HIPCHK(h, hipMalloc((void**)&h->d_fwdt[grp], sizeof(PipeFwd)));
    PipeFwd& P = *h->fwdt_host[grp];
    memset(&P, 0, sizeof(P));
    int n_nets = 0;
    auto add = [&](int ch, int kind) {
      for (int t = 0; t < 2; ++t) P.u[2 * n_nets + t] = fwd_unit_twin(h, ch, t, kind);
      return &P.u[2 * n_nets++];
    };
    if (grp == 0) {
      FwdUnit* pi = add(C_PI, HEAD_POLICY);
      for (int t = 0; t < 2; ++t) {
        pi[t].logits = h->logits_pi; pi[t].logp = h->logp_new; pi[t].eps = h->eps_new;
        pi[t].xact = h->Xc[C_Q1P]; pi[t].xact2 = nq == 2 ? h->Xc[C_Q2P] : nullptr;
      }
      pi[1].part_heads = h->part_heads;
      pi[0].x0t = h->X0t_net[2];
      FwdUnit* pt = add(C_PIT, HEAD_POLICY);
      for (int t = 0; t < 2; ++t) {
        pt[t].logits = h->logits_pit; pt[t].logp = h->logp2; pt[t].eps = h->eps_2;
        pt[t].xact = h->Xc[C_Q1T]; pt[t].xact2 = nq == 2 ? h->Xc[C_Q2T] : nullptr;
        for (int l = 0; l < L; ++l) pt[t].G[l] = nullptr;   // never differentiated
      }
      for (int i = 0; i < nq; ++i) {
        FwdUnit* qc = add(C_Q1C + i, HEAD_Q);
        for (int t = 0; t < 2; ++t) { qc[t].qout = h->qout_c[i]; qc[t].qstd = h->qstd_c[i]; }
        qc[0].x0t = h->X0t_net[i];
      }
    } else {
      for (int i = 0; i < nq; ++i) {
        FwdUnit* qt = add(C_Q1T + i, HEAD_Q);
        for (int t = 0; t < 2; ++t) {
          qt[t].qout = h->qout_t[i];
          for (int l = 0; l < L; ++l) qt[t].G[l] = nullptr;   // never differentiated
        }
      }
      for (int i = 0; i < nq; ++i) {
        FwdUnit* qp = add(C_Q1P + i, HEAD_Q);
        for (int t = 0; t < 2; ++t) qp[t].qout = h->qout_p[i];
      }
    }
    const char* name = grp == 0 ? "chain_fwd_a" : "chain_fwd_b";
    // the two trunks of a net as workgroups of their own when every (net, slice) has a ready flag: twice the workgroups
    // at half the chain length (measured at batch 256: 45 / 41 us per launch with one workgroup per net and slice)
    int rg = chain_rg(h, 2 * n_nets);
    const bool par = !h->env_twin_seq && h->B / (4 * rg) <= kChainFlagSlices;
    if (!par) rg = chain_rg(h, n_nets);
    if (grp == 0) h->twin_par = par;
    for (int i = 0; i < n_nets; ++i) { P.u[2 * i].qout = nullptr; P.u[2 * i].qstd = nullptr; }   // first trunks: no head of their own
    if (par)
      for (int i = 0; i < n_nets; ++i) {
        FwdUnit& f = P.u[2 * i];
        FwdUnit& g = P.u[2 * i + 1];
        int* flags = h->chain_flags + (4 * grp + i) * kChainFlagSlices;
        f.qout = h->twin_part[4 * grp + i]; f.qstd = nullptr; f.done = flags;
        g.zinit = h->twin_part[4 * grp + i]; g.wait0 = flags; g.wait_rows0 = 4 * rg; g.late_wait = HW_HEAD;
      }
    FwdArgs& c = P.c;
    c.n_units = 2 * n_nets;
    for (int k = 0; k < c.n_units; ++k) { P.u[k].rg = (short)rg; P.u[k].n_slices = (short)(h->B / (4 * rg)); }
    c.B = h->B; c.F = h->F; c.A = h->A; c.L = L; c.ldx = h->ldx;
    c.s_obs = h->s_obs; c.s_act = h->s_act; c.v1_stats = nq == 1; c.Cb = h->B / 16;
    c.act_scale = h->act_scale; c.act_center = h->act_center; c.lo_ls = h->cfg.min_log_std; c.hi_ls = h->cfg.max_log_std;
    c.timeline = tl_for(h, name);
    c.spin_timeout = h->handoff_dev;
    c.debug_withhold = h->debug_withhold == 1;   // tests: the first trunk of the policy never raises slice 0's flag
    const int n_slices = h->B / (4 * rg);
    if (par) {
      // XCDs 0-3 run the first trunks, 4-7 the second ones; the nets share each half (rep XCDs per net, slices dealt
      // round-robin over them). A second-trunk block sits 4 ids behind its partner: dispatched later, never before it
      const int rep = n_nets >= 3 ? 1 : 4 / n_nets, rounds = (n_slices + rep - 1) / rep;
      P.n_blocks = 8 * rounds;
      if (P.n_blocks > kPipeMaxBlocks) return fail(h, DSACT_E_INVALID, "twin forward table: %d blocks exceed %d", P.n_blocks, kPipeMaxBlocks);
      for (int b = 0; b < P.n_blocks; ++b) {
        const int x = b & 3, t = (b >> 2) & 1, round = b >> 3;
        const int net = x % n_nets, k = x / n_nets, slice = k + rep * round;
        P.blk[b] = (x < n_nets * rep && slice < n_slices) ? (((2 * net + t) << 16) | slice) : -1;
      }
      HIPCHK(h, hipMemcpy(h->d_fwdt[grp], &P, sizeof(PipeFwd), hipMemcpyHostToDevice));
      continue;
    }
    // block -> (net, slice): the slices of a net on the same XCD(s) like xcd_map_uniform
    const XcdMap map = xcd_map_uniform(n_nets);
    int rounds = 0;
    for (int x = 0; x < 8; ++x) {
      if (map.unit[x] < 0) continue;
      const int n = n_slices - map.base[x];
      const int r = n > 0 ? (n + map.stride[x] - 1) / map.stride[x] : 0;
      rounds = r > rounds ? r : rounds;
    }
    P.n_blocks = 8 * rounds;
    if (P.n_blocks > kPipeMaxBlocks) return fail(h, DSACT_E_INVALID, "twin forward table: %d blocks exceed %d", P.n_blocks, kPipeMaxBlocks);
    for (int b = 0; b < P.n_blocks; ++b) {
      const int x = b & 7, unit = map.unit[x];
      const int slice = unit < 0 ? -1 : map.base[x] + map.stride[x] * (b >> 3);
      P.blk[b] = (unit >= 0 && slice < n_slices) ? (((2 * unit) << 16) | slice) : -1;
    }
    HIPCHK(h, hipMemcpy(h->d_fwdt[grp], &P, sizeof(PipeFwd), hipMemcpyHostToDevice));
  }
  // Groups A and B in ONE launch (parallel trunks, every workgroup resident: 2 per CU): group B's units compute their
  // observation segment while the policies run, then wait (HW_LATE) for the done flags of policy / policy_target's second
  // trunk -- whose head wrote the sampled action into their input rows -- before the action segment. Every wait targets
  // blocks with lower ids (A before B, first trunks before their partners): the bounded spins cannot deadlock.
  h->twin_merged = false;
  PipeFwd& A = *h->fwdt_host[0];
  const PipeFwd& Bt = *h->fwdt_host[1];
  if (h->twin_par && getenv("DSACT_TWIN_NO_MERGE") == nullptr && A.c.n_units + Bt.c.n_units <= kPipeUnits &&
      A.n_blocks + Bt.n_blocks <= kPipeMaxBlocks && A.u[0].rg == Bt.u[0].rg) {
    const int uoff = A.c.n_units, rg = A.u[0].rg;
    int* done_pi = h->chain_flags + 8 * kChainFlagSlices;
    int* done_pit = h->chain_flags + 9 * kChainFlagSlices;
    A.u[1].done = done_pi; A.u[3].done = done_pit;     // second trunks of policy / policy_target (units 0..3: pi, pit)
    for (int k = 0; k < Bt.c.n_units; ++k) {
      FwdUnit u = Bt.u[k];
      const bool target_chain = k < 2 * nq;             // group B: q_t x nq (read act2: policy_target), then q(obs, new_act) x nq
      u.wait1 = target_chain ? done_pit : done_pi; u.wait_rows1 = 4 * rg; u.late_wait |= HW_LATE;
      A.u[uoff + k] = u;
    }
    for (int b = 0; b < Bt.n_blocks; ++b) A.blk[A.n_blocks + b] = Bt.blk[b] < 0 ? -1 : Bt.blk[b] + (uoff << 16);
    A.n_blocks += Bt.n_blocks;
    A.c.n_units += Bt.c.n_units;
    A.c.timeline = tl_for(h, "chain_fwd");
    HIPCHK(h, hipMemcpy(h->d_fwdt[0], &A, sizeof(PipeFwd), hipMemcpyHostToDevice));
    h->twin_merged = true;
  }
  return DSACT_OK;
}

int enqueue_chain_fwd_twin(dsact_handle* h, int grp) {
  if (grp == 1 && h->twin_merged) return DSACT_OK;
  const PipeFwd& P = *h->fwdt_host[grp];
  const int rg = P.u[0].rg;
  const size_t lds = (size_t)chain_lds(4 * (h->s_obs + h->s_act), h->cW, 4 * rg).total * sizeof(float);
  if (grp == 0) h->n_heads_parts = h->B / 4;   // one partial per four rows whatever the rows per workgroup (chain_fwd_body)
  if (grp == 0 && h->twin_par) {
    if (h->flags_dirty) HIPCHK(h, hipMemsetAsync(h->chain_flags, 0, kChainFlags * sizeof(int), h->stream));
    h->flags_dirty = true;
  }
  const char* name = h->twin_merged ? "chain_fwd" : grp == 0 ? "chain_fwd_a" : "chain_fwd_b";
  const PipeFwd* dev = h->d_fwdt[grp];
#define CALL_FT(N) return generic_act(h) ? launch(h, name, k_chain_fwdt<N, true>, dim3(P.n_blocks), dim3(64 * N), lds, dev) \
                                         : launch(h, name, k_chain_fwdt<N>, dim3(P.n_blocks), dim3(64 * N), lds, dev)
  if (h->cNT == 1) CALL_FT(1);
  if (h->cNT == 2) CALL_FT(2);
  CALL_FT(4);
#undef CALL_FT
}

// fat mode: 32-row workgroups when even those fill the chip twice over, else 16-row ones
int fat_rt(const dsact_handle* h, int n_units) {
  if (h->env_fat_rt) return h->env_fat_rt;
  return (n_units * (h->B / 32) >= 512 && h->B % 32 == 0) ? 2 : 1;
}

#define FAT_NT(CALL, RTV)                      \
  do {                                         \
    if ((RTV) == 2) {                          \
      if (h->cNT == 2) { CALL(2, 2); }         \
      else { CALL(4, 2); }                     \
    } else {                                   \
      if (h->cNT == 2) { CALL(2, 1); }         \
      else { CALL(4, 1); }                     \
    }                                          \
  } while (0)

int launch_chain_fwd(dsact_handle* h, const char* name, FwdArgs& a) {
  if (h->fat) {
    const int rt = fat_rt(h, a.n_units);
    fill_fwd_common(h, a, 4 * rt, name);
    // the slice's input rows staged in LDS (round 6) where they fit the default 64 KB of dynamic LDS; else (very wide
    // observations) the first layer reads them from global memory as in rounds 3-5. DSACT_NO_FAT_STAGE: the A/B switch
    size_t lds = (size_t)fat_lds(h->cW, 16 * rt, 0, 16 * (a.s_obs + a.s_act)).total * sizeof(float);
    a.x0_lds = !h->env_no_fat_stage && lds <= 64 * 1024;
    if (!a.x0_lds) lds = (size_t)fat_lds(h->cW, 16 * rt, 0).total * sizeof(float);
    const int grid = a.n_units * a.u[0].n_slices;
    if (a.u[0].part_heads) h->n_heads_parts = a.u[0].n_slices;
#define CALL_FF(N, T) return launch(h, name, k_fat_fwd<N, T>, dim3(grid), dim3(64 * N), lds, a)
#define CALL_FFG(N, T) return launch(h, name, k_fat_fwd<N, T, true>, dim3(grid), dim3(64 * N), lds, a)
    if (generic_act(h)) FAT_NT(CALL_FFG, rt);
    FAT_NT(CALL_FF, rt);
#undef CALL_FF
#undef CALL_FFG
  }
  const int rg = chain_rg(h, a.n_units);
  fill_fwd_common(h, a, rg, name);
  const int grid = fwd_grid(a);
  const size_t lds = (size_t)chain_lds(4 * (h->s_obs + h->s_act), h->cW, 4 * rg).total * sizeof(float);
  if (a.u[0].part_heads) h->n_heads_parts = h->B / 4;   // one partial per four rows whatever the rows per workgroup (chain_fwd_body)
#define CALL_CF(N, G) return launch(h, name, k_chain_fwd<N, G>, dim3(grid), dim3(64 * N), lds, a)
#define CALL_CFG(N, G) return launch(h, name, k_chain_fwd<N, G, true>, dim3(grid), dim3(64 * N), lds, a)
  if (generic_act(h)) CHAIN_NT(CALL_CFG, rg);
  CHAIN_NT(CALL_CF, rg);
#undef CALL_CF
#undef CALL_CFG
}

int enqueue_chain_fwd_a(dsact_handle* h) {
  FwdArgs a;
  fwd_args_a(h, a);
  return launch_chain_fwd(h, "chain_fwd_a", a);
}

int enqueue_chain_fwd_b(dsact_handle* h) {
  FwdArgs a;
  fwd_args_b(h, a);
  return launch_chain_fwd(h, "chain_fwd_b", a);
}

// A and B in ONE launch (batch <= 256: every workgroup of both groups fits on the chip at once): group B's workgroups
// wait for the ready flags of the group-A slices they read (k_chain_fwd2); k_chain_bwd_q clears the flags.
int enqueue_chain_fwd_merged(dsact_handle* h) {
  Fwd2Args m;
  fwd_args_a(h, m.A);
  fwd_args_b(h, m.B);
  const int rga = chain_rg(h, m.A.n_units), rgb = chain_rg(h, m.B.n_units);
  // Group A at batch 256 with 8-row workgroups is 192 workgroups on 6 XCDs, and the policy chain -- whose sampled action
  // every q(obs, new_act) consumer waits for -- is the critical path. Mixed row counts: the two policy units run 4-row
  // workgroups (0.75x the layer time) on two XCDs each, the four critic units 8-row ones on one XCD each: 256
  // workgroups, one per CU, 8 XCDs busy; with group B's 256 that is exactly two per CU.
  XcdMap mixed_map;
  const int nq = h->nq;
  const bool mixed = rga == 2 && h->B % 8 == 0 && 2 * (h->B / 4) + 2 * nq * (h->B / 8) <= 256;
  if (mixed) {
    m.A.u[0].rg = m.A.u[1].rg = 1;
    const int share6[6] = {2, 2, 1, 1, 1, 1}, share4[4] = {2, 2, 2, 2};
    mixed_map = nq == 2 ? xcd_map_shares(6, share6) : xcd_map_shares(4, share4);
  }
  fill_fwd_common(h, m.A, rga, "chain_fwd", mixed ? &mixed_map : nullptr);
  fill_fwd_common(h, m.B, rgb, "chain_fwd");
  h->n_heads_parts = h->B / 4;   // one partial per four rows whatever the rows per workgroup (chain_fwd_body)
  int* f = h->chain_flags;
  for (int k = 0; k < m.A.n_units; ++k) m.A.u[k].done = f + k * kChainFlagSlices;   // pi, pit, q_c x nq, q_t(obs) x nq
  for (int i = 0; i < nq; ++i) {
    m.A.u[2 + i].zdone = f + (6 + i) * kChainFlagSlices;   // q_c: the saved observation part is ready
    FwdUnit& qt = m.B.u[i];       // q_t(obs2, act2): action from pit, observation part from the obs-only unit
    qt.wait0 = m.A.u[1].done; qt.wait_rows0 = 4 * m.A.u[1].rg;
    qt.wait1 = m.A.u[2 + nq + i].done; qt.wait_rows1 = 4 * m.A.u[2 + nq + i].rg;
    FwdUnit& qp = m.B.u[nq + i];  // q(obs, new_act): action from pi, observation part from q_c (its early flag)
    qp.wait0 = m.A.u[0].done; qp.wait_rows0 = 4 * m.A.u[0].rg;
    qp.wait1 = m.A.u[2 + i].zdone; qp.wait_rows1 = 4 * m.A.u[2 + i].rg;
  }
  m.n_a = fwd_grid(m.A);
  const int grid = m.n_a + fwd_grid(m.B);
  // The ready flags are cleared by the critics' backward, which follows every forward of a complete update. A forward
  // that follows another forward (dsact_dp_enqueue_forward twice, or one never followed by its backward) would find them
  // raised and its consumers would not wait: clear them first (a memset node when captured).
  if (h->flags_dirty) HIPCHK(h, hipMemsetAsync(h->chain_flags, 0, kChainFlags * sizeof(int), h->stream));
  h->flags_dirty = true;
  const size_t lds = (size_t)chain_lds(4 * (h->s_obs + h->s_act), h->cW, 4 * (rga > rgb ? rga : rgb)).total * sizeof(float);
  if (generic_act(h)) {
    if (h->cNT == 1) return launch(h, "chain_fwd", k_chain_fwd2<1, true>, dim3(grid), dim3(64), lds, m);
    if (h->cNT == 2) return launch(h, "chain_fwd", k_chain_fwd2<2, true>, dim3(grid), dim3(128), lds, m);
    return launch(h, "chain_fwd", k_chain_fwd2<4, true>, dim3(grid), dim3(256), lds, m);
  }
  if (h->cNT == 1) return launch(h, "chain_fwd", k_chain_fwd2<1>, dim3(grid), dim3(64), lds, m);
  if (h->cNT == 2) return launch(h, "chain_fwd", k_chain_fwd2<2>, dim3(grid), dim3(128), lds, m);
  return launch(h, "chain_fwd", k_chain_fwd2<4>, dim3(grid), dim3(256), lds, m);
}

// ---- pipelined graph (k_chain_fwdp, dsact_chain.h) ---------------------------------------------------------------
// Per-minibatch buffers exist kPipeSets times: update s of a captured graph of n updates works on set (s - (n-1)) & 3 (the
// LAST update on set 0 = the workspace's own buffers, so whatever reads the handle after a replay -- statistics, debug
// reads, dsact_read_batch -- sees the last update as after eager updates), update s's forward launch may also fill set
// s + 1 (policy units of the next minibatch), and the gather riding in update s's critic-backward launch fills set s + 2.
int alloc_pipe_sets(dsact_handle* h) {
  if (h->pipe_ws) return DSACT_OK;
  const size_t B = h->B;
  const int A = h->A, L = h->L;
  for (int pass = 0; pass < 2; ++pass) {
    Carver c;
    c.base = pass ? h->pipe_ws : nullptr;
    for (int k = 1; k < dsact_handle::kPipeSets; ++k) {
      dsact_handle::PipeSet& p = h->pset[k];
      p.X0 = c.take<float>(B * h->ldx); p.XP = c.take<float>(B * h->ldx); p.X2 = c.take<float>(B * h->ldx);
      p.rew = c.take<float>(B); p.done = c.take<float>(B);
      p.eps_new = c.take<float>(B * A); p.eps_2 = c.take<float>(B * A); p.z5 = c.take<float>(B); p.z6 = c.take<float>(B);
      p.logits_pi = c.take<float>(B * 2 * A); p.logits_pit = c.take<float>(B * 2 * A);
      p.logp_new = c.take<float>(B); p.logp2 = c.take<float>(B);
      for (int l = 0; l < L; ++l) { p.Hpi[l] = c.take<float>(B * h->w[l]); p.Gpi[l] = c.take<float>(B * h->w[l]); }
      for (int i = 0; i < 2; ++i) p.qout_t[i] = c.take<float>(B * 2);
      p.part_heads = c.take<float>((size_t)h->n_heads_wg * 2);
      p.X0t = c.take<float>(B * (size_t)((h->F + A + 31) / 32 * 32));
    }
    for (int i = 0; i < 3; ++i) h->pipe_hand[i] = c.take<unsigned long long>(B * 32);
    if (!pass) {
      HIPCHK(h, hipMalloc((void**)&h->pipe_ws, c.off + 256));
      HIPCHK(h, hipMemset(h->pipe_ws, 0, c.off + 256));
    }
  }
  dsact_handle::PipeSet& p0 = h->pset[0];
  p0.X0 = h->X0; p0.XP = h->XP; p0.X2 = h->X2; p0.rew = h->rew; p0.done = h->done;
  p0.eps_new = h->eps_new; p0.eps_2 = h->eps_2; p0.z5 = h->z5; p0.z6 = h->z6;
  p0.logits_pi = h->logits_pi; p0.logits_pit = h->logits_pit; p0.logp_new = h->logp_new; p0.logp2 = h->logp2;
  for (int l = 0; l < L; ++l) { p0.Hpi[l] = h->Hb[C_PI][l]; p0.Gpi[l] = h->Gb[C_PI][l]; }
  for (int i = 0; i < 2; ++i) p0.qout_t[i] = h->qout_t[i];
  p0.part_heads = h->part_heads;
  p0.X0t = h->X0t;
  return DSACT_OK;
}

// the handle's per-minibatch pointers := set k (every args builder reads the handle); k = 0 restores the workspace's own
void apply_pipe_set(dsact_handle* h, int k) {
  const dsact_handle::PipeSet& p = h->pset[k];
  h->X0 = p.X0; h->XP = p.XP; h->X2 = p.X2; h->rew = p.rew; h->done = p.done;
  h->eps_new = p.eps_new; h->eps_2 = p.eps_2; h->z5 = p.z5; h->z6 = p.z6;
  h->logits_pi = p.logits_pi; h->logits_pit = p.logits_pit; h->logp_new = p.logp_new; h->logp2 = p.logp2;
  for (int l = 0; l < h->L; ++l) { h->Hb[C_PI][l] = p.Hpi[l]; h->Gb[C_PI][l] = p.Gpi[l]; }
  for (int i = 0; i < 2; ++i) h->qout_t[i] = p.qout_t[i];
  h->part_heads = p.part_heads;
  h->X0t = p.X0t;
  h->Xc[C_PI] = h->Xc[C_Q1C] = h->Xc[C_Q2C] = h->X0;
  h->Xc[C_PIT] = h->Xc[C_Q1T] = h->Xc[C_Q2T] = h->X2;
  h->Xc[C_Q1P] = h->Xc[C_Q2P] = h->XP;
}

const char* pipe_fwd_name(bool pre, bool do_pre) {
  return pre ? (do_pre ? "chain_fwd_q+next" : "chain_fwd_q") : (do_pre ? "chain_fwd+next" : "chain_fwd");
}

// roles of a pipelined forward launch, in dispatch-priority order: units that never wait, own minibatch then next; then
// their consumers. A unit waits only for units EARLIER in this order, and every XCD's queue is filled in this order.
enum PipeRole { PR_PI = 0, PR_PIT, PR_Q1C, PR_Q2C, PR_PIN, PR_PITN, PR_Q1P, PR_Q2P, PR_Q1T, PR_Q2T, PR_Q1TN, PR_Q2TN, PR_N };
static const char* kPipeRoleName[PR_N] = {"pi", "pit", "q1c", "q2c", "pin", "pitn", "q1p", "q2p", "q1t", "q2t", "q1tn", "q2tn"};
static_assert(PR_N <= kPipeUnits, "unit table too small");

// XCDs of a role in a launch shape (pre = the policy units of this minibatch were computed by the previous launch, do_pre =
// this launch computes the next minibatch's): a digit string, slices dealt round-robin over it. Placement is speed only.
// DSACT_PIPE_MAP="FT.pi=017:1;TF.q1t=45:2;..." overrides a role's XCDs (and, after ':', its rows per workgroup / 4).
static const char* pipe_xcds_default(bool pre, bool do_pre, int role) {
  if (!pre) {   // FF / FT: pi -> q_p is the critical path; pi alone on its CUs (its XCDs' second slots hold waiting q_p workgroups),
    // every other unit one XCD, q_t beside the next minibatch's policy units; no XCD beyond its 64 slots. Random placements
    // cost 7 - 16 us, single-unit moves around this one +-1 us (scripts/pipe_map_search.py, profiles/r04_pipe_map_search.txt)
    // pi and pi_target (their consumers q_p / q_t are the two critical paths) 4-row workgroups on two XCDs each, one per CU;
    // q_t's 4-row workgroups half beside pi_target (they resume when it exits), half beside the next minibatch's policy units
    static const char* t[PR_N] = {"01", "27", "3", "4", "5", "6", "30", "41", "52", "67", "52", "67"};
    return t[role];
  }
  if (!do_pre) {   // TF: only the fresh-critic chains, all independent: 8-row workgroups (half the L2 traffic of 4-row ones:
    // 22.2 vs 26.6 us, profiles/r04_pipe_map_search.txt), every unit on two or more XCDs (placement among those: equal)
    static const char* t[PR_N] = {"", "", "04", "15", "", "", "26", "37", "0246", "1357", "", ""};
    return t[role];
  }
  static const char* t[PR_N] = {"", "", "04", "15", "26", "37", "26", "37", "0246", "1357", "0246", "1357"};   // TT (delay_update >= 3)
  return t[role];
}

struct PipePlace { std::string xcds; int rg; };
static PipePlace pipe_place(const dsact_handle* h, bool pre, bool do_pre, int role, int rg_default) {
  PipePlace pl;
  pl.xcds = pipe_xcds_default(pre, do_pre, role);
  pl.rg = rg_default;
  if (!h->env_pipe_map.empty()) {
    const std::string key = std::string(pre ? "T" : "F") + (do_pre ? "T" : "F") + "." + kPipeRoleName[role] + "=";
    size_t at = 0;
    while ((at = h->env_pipe_map.find(key, at)) != std::string::npos) {
      if (at == 0 || h->env_pipe_map[at - 1] == ';') {
        size_t e = h->env_pipe_map.find(';', at);
        std::string v = h->env_pipe_map.substr(at + key.size(), e == std::string::npos ? std::string::npos : e - at - key.size());
        const size_t c = v.find(':');
        if (c != std::string::npos) { pl.rg = atoi(v.c_str() + c + 1) == 1 ? 1 : 2; v = v.substr(0, c); }
        std::string x;
        for (char ch : v) if (ch >= '0' && ch <= '7') x += ch;
        if (!x.empty()) pl.xcds = x;
        break;
      }
      at += key.size();
    }
  }
  if (pl.xcds.empty()) pl.xcds = "01234567";
  return pl;
}

// fills P for the forward launch of a pipelined update: own minibatch = set_own (pre: its policy units ran in the previous
// launch), next minibatch = set_next (do_pre: its policy units run here). Leaves the handle on set_own.
// bp: nullptr, or the deferred policy backward of the previous update this launch carries (its chain slices and tiles)
int pipe_fwd_build(dsact_handle* h, int set_own, int set_next, bool pre, bool do_pre, PipeFwd& P, const BwdPiArgs* bp = nullptr, bool book = false) {
  memset(&P, 0, sizeof(P));
  const bool qt_pre = false;   // (precomputing the NEXT minibatch's target units too -- DSACT_PIPE_QT, rounds 4-5 -- measured slower and removed)
  int* f = h->chain_flags;
  const int B = h->B;
  int idx[PR_N];
  for (int r = 0; r < PR_N; ++r) idx[r] = -1;
  int rgs[PR_N];
  std::string xc[PR_N];
  // rows per workgroup: the critical units (pi -> q_p) run 4-row workgroups; the others 8-row ones (39 % less CU time per
  // row) unless the launch leaves CUs idle anyway; a consumer never has more rows than its producer (it waits for ONE flag)
  const int side = (B % 8 == 0) ? 2 : 1, nxt = side;   // 8-row workgroups off the critical path (4-row ones: measured slower, rounds 4-5)
  const int dflt[PR_N] = {1, pre ? side : 1, side, side, nxt, nxt, pre ? side : 1, pre ? side : 1, pre ? side : 1, pre ? side : 1, side, side};
  for (int r = 0; r < PR_N; ++r) {
    const PipePlace pl = pipe_place(h, pre, do_pre, r, dflt[r]);
    rgs[r] = (B % 8 == 0) ? pl.rg : 1; xc[r] = pl.xcds;
  }
  auto cap = [&](int consumer, int producer) { if (rgs[consumer] > rgs[producer]) rgs[consumer] = rgs[producer]; };
  if (!pre) { cap(PR_Q1P, PR_Q1C); cap(PR_Q2P, PR_Q2C); }
  if (!pre && h->env_no_pipe_tagged) { cap(PR_Q1P, PR_PI); cap(PR_Q2P, PR_PI); cap(PR_Q1T, PR_PIT); cap(PR_Q2T, PR_PIT); }
  if (h->env_no_pipe_tagged) { cap(PR_Q1TN, PR_PITN); cap(PR_Q2TN, PR_PITN); }   // (tagged pairs are polled per element: any rows)
  auto put = [&](int role, const FwdUnit& u) {
    idx[role] = role;
    P.u[role] = u;
    P.u[role].rg = (short)rgs[role];
    P.u[role].n_slices = (short)(B / (4 * rgs[role]));
  };
  // in-launch hand-over of the sampled actions: (value, tag) pairs (the data is the flag) or, DSACT_NO_PIPE_TAGGED, ready flags
  const bool tagged = !h->env_no_pipe_tagged;
  auto policy_units = [&](int r_pi, int r_pit, bool flag_pi, bool flag_pit) {
    FwdUnit pi = fwd_unit(h, C_PI, SEG_FULL, HEAD_POLICY);
    pi.logits = h->logits_pi; pi.logp = h->logp_new; pi.eps = h->eps_new; pi.xact = h->Xc[C_Q1P]; pi.part_heads = h->part_heads;
    if (flag_pi && tagged) { pi.xact2 = (float*)h->pipe_hand[0]; pi.late_wait |= HW_PAIRS_OUT; }
    else if (flag_pi) pi.done = f + 0 * kChainFlagSlices;
    put(r_pi, pi);
    FwdUnit pt = fwd_unit(h, C_PIT, SEG_FULL, HEAD_POLICY);
    pt.logits = h->logits_pit; pt.logp = h->logp2; pt.eps = h->eps_2; pt.xact = h->Xc[C_Q1T];
    for (int l = 0; l < h->L; ++l) pt.G[l] = nullptr;   // never differentiated
    if (flag_pit && tagged) { pt.xact2 = (float*)h->pipe_hand[r_pit == PR_PIT ? 1 : 2]; pt.late_wait |= HW_PAIRS_OUT; }
    else if (flag_pit) pt.done = f + (r_pit == PR_PIT ? 1 : 4) * kChainFlagSlices;
    put(r_pit, pt);
  };
  auto target_units = [&](int r_q1t, int r_pit, bool wait) {
    for (int i = 0; i < h->nq; ++i) {   // (one critic: DSAC_V1)
      FwdUnit qt = fwd_unit(h, C_Q1T + i, SEG_FULL_SPLIT, HEAD_Q);
      qt.qout = h->qout_t[i];
      for (int l = 0; l < h->L; ++l) qt.G[l] = nullptr;   // never differentiated
      if (wait && tagged) { qt.wait0 = (const int*)h->pipe_hand[r_pit == PR_PIT ? 1 : 2]; qt.wait_rows0 = 4; qt.late_wait = HW_LATE | HW_PAIRS_IN; }
      else if (wait) { qt.wait0 = P.u[r_pit].done; qt.wait_rows0 = 4 * rgs[r_pit]; qt.late_wait = HW_LATE; }
      put(r_q1t + i, qt);
    }
  };
  apply_pipe_set(h, set_own);
  if (!pre) policy_units(PR_PI, PR_PIT, true, true);
  // q(obs, new_act): eager updates run it as "saved observation part of q(obs, act)'s first layer + action part". A unit
  // that computes the observation part itself and merges the accumulators where the hand-over would have (SEG_FULL_SPLIT)
  // produces the same bits with no producer: that is how the launches whose policy units ran earlier hold it (no in-launch
  // dependency at all), and -- DSACT_PIPE_QP_SPLIT=1 -- optionally the others (observation part under the wait for pi)
  const bool qp_split = pre;
  for (int i = 0; i < h->nq; ++i) {   // (one critic: DSAC_V1)
    FwdUnit qc = fwd_unit(h, C_Q1C + i, qp_split ? SEG_FULL : SEG_FULL_SAVE, HEAD_Q);
    qc.qout = h->qout_c[i]; qc.qstd = h->qstd_c[i];
    if (i == 0) qc.x0t = h->X0t;
    if (!qp_split) { qc.zsave = h->zobs[i]; qc.zdone = f + (2 + i) * kChainFlagSlices; }
    put(PR_Q1C + i, qc);
    FwdUnit qp = fwd_unit(h, C_Q1P + i, qp_split ? SEG_FULL_SPLIT : SEG_ACT_FROM_SAVED, HEAD_Q);
    qp.qout = h->qout_p[i];
    if (!pre && tagged) { qp.wait0 = (const int*)h->pipe_hand[0]; qp.wait_rows0 = 4; qp.late_wait = HW_LATE | HW_PAIRS_IN; }
    else if (!pre) { qp.wait0 = P.u[PR_PI].done; qp.wait_rows0 = 4 * rgs[PR_PI]; qp.late_wait = HW_LATE; }
    if (!qp_split) { qp.zinit = h->zobs[i]; qp.wait1 = qc.zdone; qp.wait_rows1 = 4 * rgs[PR_Q1C + i]; }
    put(PR_Q1P + i, qp);
  }
  if (!pre || !qt_pre) target_units(PR_Q1T, PR_PIT, !pre);
  if (do_pre) {
    apply_pipe_set(h, set_next);
    policy_units(PR_PIN, PR_PITN, false, qt_pre);
    if (qt_pre) target_units(PR_Q1TN, PR_PITN, true);
    apply_pipe_set(h, set_own);
  }
  // common fields
  FwdArgs& a = P.c;
  a.n_units = PR_N;
  a.B = h->B; a.F = h->F; a.A = h->A; a.L = h->L; a.ldx = h->ldx;
  a.s_obs = h->s_obs; a.s_act = h->s_act; a.v1_stats = h->nq == 1; a.Cb = h->B / 16;
  a.act_scale = h->act_scale; a.act_center = h->act_center; a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std;
  a.timeline = tl_for(h, pipe_fwd_name(pre, do_pre));
  a.spin_timeout = h->handoff_dev;
  a.debug_withhold = h->debug_withhold == 1;
  a.tagp = &h->st->tag_seq;
  a.tpad = h->env_pk_pad;
  // block table: every XCD's queue is filled role by role (the enum is the priority order), a role's slices are dealt
  // round-robin over its XCDs; block 8 r + x = entry r of XCD x's queue (the dispatcher places block b on XCD b % 8)
  std::vector<int> q[8];
  for (int r = 0; r < PR_N; ++r) {
    if (idx[r] < 0) continue;
    if (bp && r == PR_Q1P)   // the deferred chain's slices: producers of the tiles at the end of every queue; they never wait
      for (int sl = 0; sl < bp->n_slices; ++sl) q[sl & 7].push_back((kPipeRoleBwdPi << 16) | sl);
    const int ns = P.u[r].n_slices;
    for (int sl = 0; sl < ns; ++sl) q[xc[r][sl % xc[r].size()] - '0'].push_back((r << 16) | sl);
  }
  if (bp) {   // the deferred policy tiles: XCD x takes the x-th contiguous eighth of the tile list (xcd_chunk's locality)
    const int per = (bp->n_pi_tiles + 7) >> 3;
    for (int t = 0; t < bp->n_pi_tiles; ++t) q[t / per].push_back((kPipeRoleTile << 16) | t);
  }
  if (book) {   // this update's bookkeeping + the reset of k_chain_bwd_qt's arrival counters: one thread, at the end of the shortest queue
    int xs = 0;
    for (int x = 1; x < 8; ++x) if (q[x].size() < q[xs].size()) xs = x;
    q[xs].push_back(kPipeRoleBook << 16);
    P.book_st = h->st; P.book_hp = step_hyper(h); P.book_cnt = h->bqt_cnt; P.book_ncnt = 3 * 8;
  }
  size_t rounds = 0;
  for (int x = 0; x < 8; ++x) rounds = q[x].size() > rounds ? q[x].size() : rounds;
  if (8 * rounds > (size_t)kPipeMaxBlocks) return fail(h, DSACT_E_INVALID, "pipelined forward: block table too small");
  P.n_blocks = (int)(8 * rounds);
  for (size_t r = 0; r < rounds; ++r)
    for (int x = 0; x < 8; ++x) P.blk[8 * r + x] = r < q[x].size() ? q[x][r] : -1;
  if (!h->env_no_pipe_warm) {   // L2 warm-up shares: (index, count) among the workgroups of the same unit on the same XCD
    for (int x = 0; x < 8; ++x) {
      int cnt[PR_N], seen[PR_N];
      for (int r = 0; r < PR_N; ++r) cnt[r] = seen[r] = 0;
      for (int code : q[x]) if ((code >> 16) < PR_N) cnt[code >> 16]++;
      for (size_t r = 0; r < q[x].size(); ++r) {
        const int role = q[x][r] >> 16;
        if (role < PR_N) P.warm[8 * r + x] = (seen[role]++ << 16) | cnt[role];
      }
    }
  }
  h->n_heads_parts = h->B / 4;
  return DSACT_OK;
}

int launch_chain_fwd_pipe(dsact_handle* h, const char* name, const PipeFwd& host, const PipeFwd* dev, const BwdPiArgs* bp = nullptr, int bp_rg = 2) {
  // (forwards of a complete update: the critics' backward clears the ready flags)
  if (h->flags_dirty) HIPCHK(h, hipMemsetAsync(h->chain_flags, 0, kChainFlags * sizeof(int), h->stream));
  h->flags_dirty = true;
  size_t lds = (size_t)chain_lds(4 * (h->s_obs + h->s_act), h->cW, 8).total * sizeof(float);
  const int grid = host.n_blocks;
  if (bp) {   // + the previous update's deferred policy backward (k_chain_fwdpb, 256 threads for its tiles)
    const size_t lb = (size_t)chain_lds(4 * h->SoT, h->cW, 4 * bp_rg).total * sizeof(float);
    if (lb > lds) lds = lb;
    if (lds < kDw2LdsFloats * sizeof(float)) lds = kDw2LdsFloats * sizeof(float);
    if (generic_act(h)) {
      if (h->cNT == 1) return launch(h, name, k_chain_fwdpb<1, true>, dim3(grid), dim3(256), lds, dev, *bp, bp_rg);
      if (h->cNT == 2) return launch(h, name, k_chain_fwdpb<2, true>, dim3(grid), dim3(256), lds, dev, *bp, bp_rg);
      return launch(h, name, k_chain_fwdpb<4, true>, dim3(grid), dim3(256), lds, dev, *bp, bp_rg);
    }
    if (h->cNT == 1) return launch(h, name, k_chain_fwdpb<1>, dim3(grid), dim3(256), lds, dev, *bp, bp_rg);
    if (h->cNT == 2) return launch(h, name, k_chain_fwdpb<2>, dim3(grid), dim3(256), lds, dev, *bp, bp_rg);
    return launch(h, name, k_chain_fwdpb<4>, dim3(grid), dim3(256), lds, dev, *bp, bp_rg);
  }
  if (generic_act(h)) {
    if (h->cNT == 1) return launch(h, name, k_chain_fwdp<1, true>, dim3(grid), dim3(64), lds, dev);
    if (h->cNT == 2) return launch(h, name, k_chain_fwdp<2, true>, dim3(grid), dim3(128), lds, dev);
    return launch(h, name, k_chain_fwdp<4, true>, dim3(grid), dim3(256), lds, dev);
  }
  if (h->cNT == 1) return launch(h, name, k_chain_fwdp<1>, dim3(grid), dim3(64), lds, dev);
  if (h->cNT == 2) return launch(h, name, k_chain_fwdp<2>, dim3(grid), dim3(128), lds, dev);
  return launch(h, name, k_chain_fwdp<4>, dim3(grid), dim3(256), lds, dev);
}

// loss + dZ chains of the critics (n_units 2: q1c, q2c only -- off iterations of the delayed update) and of
// q1/q2(obs,new_act); riders as in the loss launch of the tile path
// arguments of the critics' backward; returns the row groups per workgroup and the rider count
void bwd_q_args(dsact_handle* h, int n_units, const RideArgs* ride, BwdQArgs& a, int& rg_out, int& n_riders_out) {
  memset(&a, 0, sizeof(a));
  const int L = h->L;
  // units: the critic chains q_c x nq, then (actor backward) the chains q(obs, new_act) x nq; `which` keeps DSAC_V2's
  // numbering (0 q1c, 1 q2c, 2 q1p, 3 q2p) so that the row phase and the buffers are the same for one critic (DSAC_V1)
  const int chs[4] = {C_Q1C, C_Q2C, C_Q1P, C_Q2P};
  for (int w = 0; w < n_units; ++w) {
    BwdQUnit& u = a.u[w];
    const int which = h->nq == 2 ? w : 2 * w;
    const int ch = chs[which];
    const int net = kChainNet[ch], n3 = net == N_Q1 ? 0 : 1;
    for (int l = 1; l < L; ++l) u.wb[l] = h->pk_bwd[n3][l];
    u.wout = net_params(h, net) + h->qd.w_off[L];
    for (int l = 0; l < L; ++l) { u.G[l] = h->Gb[ch][l]; u.dZ[l] = h->dZ[kDzSlot[ch]][l]; }
    u.dout = h->dout[which];
    if (which < 2) u.doutT = h->doutT[which];
    if (which >= 2) { u.w1at = h->pk_w1at[n3]; u.dA = h->dAq[n3]; }
    u.which = which;
  }
  if (h->twin) {
    // every chain becomes two trunk units (mean, log_std) over their halves of the twin-width buffers
    const int W = h->cW, tiles = W / 64, SH = W / 4, CH = W / 16;
    const size_t tB = (size_t)W * h->B;
    for (int w = n_units - 1; w >= 0; --w) {
      const BwdQUnit src = a.u[w];
      const int which = src.which, n3 = (which & 1);
      for (int t = 1; t >= 0; --t) {
        BwdQUnit& u = a.u[2 * w + t];
        u = src;
        for (int l = 1; l < L; ++l) u.wb[l] = src.wb[l] + (size_t)t * tiles * SH * 256;
        u.wout = src.wout + (size_t)t * W;
        for (int l = 0; l < L; ++l) { u.G[l] = src.G[l] + t * tB; u.dZ[l] = src.dZ[l] + t * tB; }
        if (which >= 2) { u.w1at = src.w1at + (size_t)t * CH * 256; u.dA = h->dAq[n3 + 2 * t]; }
        else u.dz0row = h->dz0row[n3] + (size_t)t * W;
        u.trunk = t;
      }
    }
    n_units *= 2;
    a.ldo = 2 * W; a.c1at = 2 * CH; a.ldz0 = h->w[0];
  }
  a.v1 = h->nq == 1; a.td_bound = h->cfg.td_bound; a.v1_bound = h->cfg.v1_unbounded ? 0 : 1;
  a.q_out_act = h->cfg.value_out_act;
  const int rg = h->fat_bwd ? 4 * fat_rt(h, n_units) : chain_rg(h, n_units, true);
  a.n_units = n_units; a.n_slices = h->B / (4 * rg); a.B = h->B; a.A = h->A; a.L = L; a.Cb = h->B / 16;
  for (int i = 0; i < 2; ++i) {   // (one critic: the second slots repeat the first -- the DSAC_V1 row phase never reads them)
    const int k = i < h->nq ? i : 0;
    a.qout_c[i] = h->qout_c[k]; a.qstd_c[i] = h->qstd_c[k]; a.qout_t[i] = h->qout_t[k]; a.qout_p[i] = h->qout_p[k];
  }
  a.rew = h->rew; a.done = h->done; a.logp2 = h->logp2; a.logp_new = h->logp_new; a.z5 = h->z5; a.z6 = h->z6;
  a.log_alpha = h->online + h->n_online - 1;
  a.part_loss = h->part_loss; a.grads_tail = h->grads + h->n_online; a.st = h->st;
  a.inv_B = 1.0f / (float)h->B;
  a.inv_Bg = h->use_std_sums ? 1.0f / (float)h->cfg.global_batch : 1.0f / (float)h->B;
  a.std_sums = (h->use_std_sums && h->nq == 2) ? h->std_sums : nullptr;
  a.auto_alpha = h->cfg.auto_alpha; a.alpha_fixed = h->cfg.alpha_fixed; a.gamma = h->cfg.gamma; a.tau_b = h->cfg.tau_b;
  a.one_minus_tau_b = (float)(1.0 - h->cfg.tau_b);
  a.n_chain_blocks = h->fat_bwd ? n_units * a.n_slices : chain_grid(n_units, a.n_slices);
  a.timeline = tl_for(h, "chain_bwd_q");
  if (h->fwd_merge || h->twin_par) { a.flags_reset = h->chain_flags; a.n_flags = kChainFlags; h->flags_dirty = false; }
  if (h->pi_merge) { a.flags_reset = h->chain_flags; a.n_flags = kChainFlagInts; h->flags_dirty = false; }   // + the arrival counters
  if (ride) a.ride = *ride;
  a.ride.n_loss_blocks = a.n_chain_blocks;
  n_riders_out = ride ? ride->n_gather + (ride->bookkeeping ? 1 : 0) : 0;
  rg_out = rg;
}

int enqueue_chain_bwd_q(dsact_handle* h, int n_units, const RideArgs* ride) {
  BwdQArgs a;
  int rg, n_riders;
  bwd_q_args(h, n_units, ride, a, rg, n_riders);
  if (h->fat_bwd) {
    const int rt = rg / 4;
    const size_t flds = (size_t)fat_lds(h->cW, 16 * rt, 0).total * sizeof(float);
#define CALL_FQ(N, T) return launch(h, "chain_bwd_q", k_fat_bwd_q<N, T>, dim3(a.n_chain_blocks + n_riders), dim3(kThreads), flds, a)
    FAT_NT(CALL_FQ, rt);
#undef CALL_FQ
  }
  const size_t lds = (size_t)chain_lds(h->cW, h->cW, 4 * rg).total * sizeof(float);
#define CALL_CQ(N, G) return launch(h, "chain_bwd_q", k_chain_bwd_q<N, G>, dim3(a.n_chain_blocks + n_riders), dim3(kThreads), lds, a)
  CHAIN_NT(CALL_CQ, rg);
#undef CALL_CQ
}

// rsample backward + policy dZ chain; weight-gradient tiles [x0, x1) ride along on the other CUs
// merge: the policy's tiles [x1, dw2_off[3]) and (fused) the closing block ride behind the riders [x0, x1)
void bwd_pi_args(dsact_handle* h, int x0, int x1, bool fused, BwdPiArgs& a, int& rg_out, bool merge = false, int rg_force = 0) {
  memset(&a, 0, sizeof(a));
  const int L = h->L;
  a.dA[0] = h->dAq[0]; a.dA[1] = h->dAq[1];
  a.logits_pi = h->logits_pi; a.eps_new = h->eps_new; a.log_alpha = h->online + h->n_online - 1;
  a.woutT = h->pk_bwd[2][L]; a.SoT = h->fat_bwd ? h->c_out : h->SoT;
  for (int l = 1; l < L; ++l) a.wb[l] = h->pk_bwd[2][l];
  for (int l = 0; l < L; ++l) { a.G[l] = h->Gb[C_PI][l]; a.dZ[l] = h->dZ[kDzSlot[C_PI]][l]; }
  a.dout_pi = h->dout_pi; a.d_new_act = h->d_new_act; a.dout_piT = h->doutT[2];
  // the policy chain shares its launch with ~2 rounds of weight-gradient tiles, which bound it: 8-row workgroups leave
  // them 32 more CUs (measured: 15.7 us vs 16.3 us with 4-row workgroups at batch 256)
  const int rg_pi = 0;   // (4-row slices with the merged tiles waiting for the chain measured 20.99 vs 20.04 us, round 3: the chain's own choice)
  const int rg = rg_force ? rg_force : h->fat_bwd ? 4 * fat_rt(h, 1) : rg_pi ? rg_pi : h->env_chain_rg ? h->env_chain_rg : (h->B >= 8 ? h->cRG : 1);
  a.n_slices = h->B / (4 * rg); a.B = h->B; a.A = h->A; a.L = L; a.Cb = h->B / 16;
  a.inv_B = 1.0f / (float)h->B; a.auto_alpha = h->cfg.auto_alpha; a.alpha_fixed = h->cfg.alpha_fixed;
  a.act_scale = h->act_scale; a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std;
  a.pi_out_act = h->cfg.policy_out_act; a.pi_out_n = h->cfg.policy_std_param ? h->A : 2 * h->A;
  a.part_loss = h->part_loss; a.n_part = h->B; a.target_entropy = -(float)h->A;
  a.grad_log_alpha = h->grads + h->n_online - 1;
  a.n_chain_blocks = roundup(a.n_slices, 8);   // the riders' first block lands on XCD 0 (xcd_chunk)
  a.n_trunks = 1;
  if (h->twin) {
    const int W = h->cW, tiles = W / 64, SH = W / 4;
    const size_t tB = (size_t)W * h->B;
    a.n_trunks = 2;
    a.woutT1 = a.woutT + (size_t)tiles * a.SoT * 256;
    for (int l = 1; l < L; ++l) a.wb1[l] = a.wb[l] + (size_t)tiles * SH * 256;
    for (int l = 0; l < L; ++l) { a.G1[l] = a.G[l] + tB; a.dZ1[l] = a.dZ[l] + tB; }
    a.dA2[0] = h->dAq[2]; a.dA2[1] = h->dAq[3];
    a.dz0row = h->dz0row[2]; a.ldz0 = h->w[0];
    a.n_chain_blocks *= 2;
  }
  a.timeline = tl_for(h, "chain_bwd_pi");
  a.dw = dw2_args(h, fused);
  a.tile0 = x0; a.n_extra = x1 > x0 ? x1 - x0 : 0;
  if (merge) {
    a.merge_dw = 1; a.pi_tile0 = x1; a.n_pi_tiles = h->dw2_off[3] - x1; a.finalize = fused ? 1 : 0;
    a.cnt_pi = h->chain_flags + kChainFlags + 128;
    a.spin_timeout = (int*)h->handoff_dev;
    a.debug_withhold = h->debug_withhold == 2;
  }
  rg_out = rg;
}

int enqueue_chain_bwd_pi(dsact_handle* h, int x0, int x1, bool fused, bool merge = false) {
  BwdPiArgs a;
  int rg;
  bwd_pi_args(h, x0, x1, fused, a, rg, merge);
  if (h->fat_bwd) {
    const int rt = rg / 4;
    size_t flds = (size_t)fat_lds(h->cW, 16 * rt, 16 * h->c_out).total * sizeof(float);
    if (flds < kDw2LdsFloats * sizeof(float)) flds = kDw2LdsFloats * sizeof(float);
#define CALL_FP(N, T) return launch(h, "chain_bwd_pi", k_fat_bwd_pi<N, T>, dim3(a.n_chain_blocks + xcd_chunk_grid(a.n_extra) * h->dw_chunks), dim3(kThreads), flds, a)
    FAT_NT(CALL_FP, rt);
#undef CALL_FP
  }
  size_t lds = (size_t)chain_lds(4 * h->SoT, h->cW, 4 * rg).total * sizeof(float);
  if (lds < kDw2LdsFloats * sizeof(float)) lds = kDw2LdsFloats * sizeof(float);
  const int tail = merge ? xcd_chunk_grid(a.n_extra) + xcd_chunk_grid(a.n_pi_tiles) + 1 : xcd_chunk_grid(a.n_extra) * h->dw_chunks;
#define CALL_CP(N, G) return launch(h, "chain_bwd_pi", k_chain_bwd_pi<N, G>, dim3(a.n_chain_blocks + tail), dim3(kThreads), lds, a)
  CHAIN_NT(CALL_CP, rg);
#undef CALL_CP
}

// pipelined graph, update whose policy backward is deferred into the next forward launch (k_chain_fwdpb): the last launch
// of the update holds the critics' weight-gradient / Adam tiles and the block that closes the update, nothing else
int enqueue_chain_bwd_pi_close(dsact_handle* h, bool fused) {
  BwdPiArgs a;
  int rg;
  bwd_pi_args(h, h->dw2_off[0], h->dw2_off[2], fused, a, rg, true);
  a.n_slices = 0; a.n_chain_blocks = 0; a.n_pi_tiles = 0;   // no chain slice arrives, nobody but the closing block waits (for 0 arrivals)
  const size_t lds = kDw2LdsFloats * sizeof(float);
  const int grid = xcd_chunk_grid(a.n_extra) + 1;
#define CALL_CPC(N, G) return launch(h, "chain_dw_q", k_chain_bwd_pi<N, G>, dim3(grid), dim3(kThreads), lds, a)
  CHAIN_NT(CALL_CPC, 2);
#undef CALL_CPC
}

// k_chain_bwd_qt (dsact_chain.h): may the critics' backward, their weight-gradient tiles and the closing block of an update
// that defers its policy backward be ONE launch?
bool bqt_ok(const dsact_handle* h) {
  return h->chain_ok && h->pi_merge && h->dw_chunks == 1 && !h->fat_bwd && !h->twin && !h->cnn && !h->env_no_bqt;
}
// counters + the block -> tile table: tiles class by class in the order their operands arrive (output + last hidden layer,
// ..., first layer), every class dealt to the XCDs in contiguous eighths (xcd_chunk's locality), block 8 r + x = entry r of XCD x
int build_bqt(dsact_handle* h) {
  if (h->bqt_tab) return DSACT_OK;
  HIPCHK(h, hipMalloc((void**)&h->bqt_cnt, kBqtCntInts * sizeof(int)));
  HIPCHK(h, hipMemset(h->bqt_cnt, 0, kBqtCntInts * sizeof(int)));
  for (int i = 0; i < 2; ++i) {
    HIPCHK(h, hipMalloc((void**)&h->bqp_pairs[i], (size_t)h->B * 32 * sizeof(unsigned long long)));
    HIPCHK(h, hipMemset(h->bqp_pairs[i], 0, (size_t)h->B * 32 * sizeof(unsigned long long)));
  }
  const Dw2Args d = dw2_args(h, true);
  const int L = h->L;
  std::vector<int> q[8];
  for (int c = L - 1; c >= 0; --c) {
    std::vector<int> cls;
    for (int net = 0; net < h->nq; ++net)
      for (int l = 0; l <= L; ++l) {
        if ((l < L ? l : L - 1) != c) continue;
        const int pi = net * (L + 1) + l;
        for (int t = pi ? d.p[pi - 1].tile_end : 0; t < d.p[pi].tile_end; ++t) cls.push_back(t);
      }
    const int per = ((int)cls.size() + 7) >> 3;
    size_t depth = 0;
    for (int x = 0; x < 8; ++x) depth = q[x].size() > depth ? q[x].size() : depth;
    for (int x = 0; x < 8; ++x) q[x].resize(depth, -1);      // a class starts at the same depth on every XCD
    for (size_t i = 0; i < cls.size(); ++i) q[i / (size_t)per].push_back(cls[i]);
  }
  size_t rounds = 0;
  for (int x = 0; x < 8; ++x) rounds = q[x].size() > rounds ? q[x].size() : rounds;
  std::vector<int> tab(8 * rounds, -1);
  for (size_t r = 0; r < rounds; ++r)
    for (int x = 0; x < 8; ++x) if (r < q[x].size()) tab[8 * r + x] = q[x][r];
  HIPCHK(h, hipMalloc((void**)&h->bqt_tab, tab.size() * sizeof(int)));
  HIPCHK(h, hipMemcpy(h->bqt_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
  h->bqt_tab_blocks = (int)tab.size();
  return DSACT_OK;
}
// the 4-unit form of the critics' backward arguments (the merged launches' kernel-argument budget)
static void shrink_bwd_q(const BwdQArgs& s8, BwdQArgsN<4>& d) {
  (BwdQTail&)d = (const BwdQTail&)s8;
  for (int i = 0; i < 4; ++i) d.u[i] = s8.u[i];
}
int enqueue_chain_bwd_qt(dsact_handle* h, bool fused, const RideArgs* ride) {
  BwdQtArgs a;
  memset(&a, 0, sizeof(a));
  int rg, n_riders;
  BwdQArgs q8;
  bwd_q_args(h, 2 * h->nq, ride, q8, rg, n_riders);
  shrink_bwd_q(q8, a.q);
  a.q.arrive = h->bqt_cnt;
  a.q.debug_withhold = h->debug_withhold == 3;
  a.q.timeline = tl_for(h, "chain_bwd_qt");
  a.dw = dw2_args(h, fused);
  a.tile_tab = h->bqt_tab; a.n_tile_blocks = h->bqt_tab_blocks;
  a.n_riders = n_riders;
  a.need = a.q.n_slices;
  a.spin_timeout = h->handoff_dev;
  a.logp_new = h->logp_new; a.n_part = h->B; a.target_entropy = -(float)h->A;
  a.grad_log_alpha = h->grads + h->n_online - 1;
  a.finalize = fused ? 1 : 0;
  size_t lds = (size_t)chain_lds(h->cW, h->cW, 4 * rg).total * sizeof(float);
  if (lds < kDw2LdsFloats * sizeof(float)) lds = kDw2LdsFloats * sizeof(float);
  const int grid = a.q.n_chain_blocks + n_riders + a.n_tile_blocks + 1;
  // (4- and 8-row slices only: the 16-row instantiations spilled 20-92 B under this kernel's three-workgroups-per-CU register
  //  bound and were reachable only where bqt_ok() is false -- batch >= 1024, pi_merge off)
  if (rg != 1 && rg != 2) return fail(h, DSACT_E_STATE, "k_chain_bwd_qt expects 4- or 8-row critic slices");
#define CALL_CQT(N, G) return launch(h, "chain_bwd_qt", k_chain_bwd_qt<N, G>, dim3(grid), dim3(kThreads), lds, a)
  if (rg == 1) { if (h->cNT == 1) { CALL_CQT(1, 1); } else if (h->cNT == 2) { CALL_CQT(2, 1); } else { CALL_CQT(4, 1); } }
  if (h->cNT == 1) { CALL_CQT(1, 2); } else if (h->cNT == 2) { CALL_CQT(2, 2); }
  CALL_CQT(4, 2);
#undef CALL_CQT
}

// k_chain_bwd_qpt: critics' backward -> policy backward -> every weight-gradient tile -> close, one launch
int enqueue_chain_bwd_qpt(dsact_handle* h, bool fused, const RideArgs* ride) {
  BwdQpArgs a;
  memset(&a, 0, sizeof(a));
  int rg_q, n_riders, rg_pi;
  BwdQArgs q8;
  bwd_q_args(h, 2 * h->nq, ride, q8, rg_q, n_riders);
  shrink_bwd_q(q8, a.q);
  a.q.arrive = h->bqt_cnt;
  a.q.debug_withhold = h->debug_withhold == 3;
  a.q.tagp = &h->st->tag_seq;
  for (int w = 0; w < 4; ++w)
    if (a.q.u[w].which >= 2) a.q.u[w].dA_pairs = h->bqp_pairs[a.q.u[w].which - 2];
  bwd_pi_args(h, h->dw2_off[0], h->dw2_off[2], fused, a.pi, rg_pi, true);
  a.pi.cnt_pi = h->bqt_cnt + 2 * 8 * kArriveStride;    // (its own counter: the forward launch's deferred chain uses the flags' one)
  a.pi.dA_pairs[0] = h->bqp_pairs[0]; a.pi.dA_pairs[1] = h->bqp_pairs[1];
  a.pi.tagp = &h->st->tag_seq;
  a.q.timeline = tl_for(h, "chain_bwd_qpt");
  a.pi.timeline = a.q.timeline;
  a.q.ride.n_loss_blocks = a.q.n_chain_blocks + a.pi.n_chain_blocks;   // loss_rider numbers its blocks from the first rider
  a.tile_tab = h->bqt_tab; a.n_tile_blocks = h->bqt_tab_blocks;
  a.n_riders = n_riders;
  a.need_c = a.q.n_slices;
  a.spin_timeout = h->handoff_dev;
  a.logp_new = h->logp_new; a.n_part = h->B; a.target_entropy = -(float)h->A;
  a.grad_log_alpha = h->grads + h->n_online - 1;
  a.finalize = fused ? 1 : 0;
  size_t lds = (size_t)chain_lds(h->cW, h->cW, 4 * rg_q).total * sizeof(float);
  const size_t lp = (size_t)chain_lds(4 * h->SoT, h->cW, 4 * rg_pi).total * sizeof(float);
  if (lp > lds) lds = lp;
  if (lds < kDw2LdsFloats * sizeof(float)) lds = kDw2LdsFloats * sizeof(float);
  const int grid = a.q.n_chain_blocks + a.pi.n_chain_blocks + n_riders + a.n_tile_blocks + xcd_chunk_grid(a.pi.n_pi_tiles) + 1;
  if (rg_q != 1 || (rg_pi != 2 && rg_pi != 1)) return fail(h, DSACT_E_STATE, "k_chain_bwd_qpt expects 4-row critic slices and 4- or 8-row policy slices");
  if (rg_pi == 1) {
    if (h->cNT == 1) return launch(h, "chain_bwd_qpt", k_chain_bwd_qpt<1, 1, 1>, dim3(grid), dim3(kThreads), lds, a);
    if (h->cNT == 2) return launch(h, "chain_bwd_qpt", k_chain_bwd_qpt<2, 1, 1>, dim3(grid), dim3(kThreads), lds, a);
    return launch(h, "chain_bwd_qpt", k_chain_bwd_qpt<4, 1, 1>, dim3(grid), dim3(kThreads), lds, a);
  }
  if (h->cNT == 1) return launch(h, "chain_bwd_qpt", k_chain_bwd_qpt<1, 1, 2>, dim3(grid), dim3(kThreads), lds, a);
  if (h->cNT == 2) return launch(h, "chain_bwd_qpt", k_chain_bwd_qpt<2, 1, 2>, dim3(grid), dim3(kThreads), lds, a);
  return launch(h, "chain_bwd_qpt", k_chain_bwd_qpt<4, 1, 2>, dim3(grid), dim3(kThreads), lds, a);
}

// same contract as enqueue_grads (phases, fused optimiser, riders of the loss launch)
int enqueue_grads_chain(dsact_handle* h, bool actor_backward, bool fused, int phase, const RideArgs* ride) {
  const int* off = h->dw2_off;
  const bool one_dfeat = h->twin && actor_backward && (phase == 0 || phase == 2) && h->dw_chunks == 1;
  if (phase == 4) goto actor_part;
  if (phase != 2) {
    if (h->cnn) TRY(enqueue_conv_forward(h));
    if (h->twin) {
      TRY(enqueue_chain_fwd_twin(h, 0));
      TRY(enqueue_chain_fwd_twin(h, 1));
    } else if (h->fwd_merge) {
      TRY(enqueue_chain_fwd_merged(h));
    } else {
      TRY(enqueue_chain_fwd_a(h));
      TRY(enqueue_chain_fwd_b(h));
    }
    if (h->use_std_sums && h->nq == 2) {
      StdSumArgs s;
      s.qstd_c[0] = h->qstd_c[0]; s.qstd_c[1] = h->qstd_c[1]; s.B = h->B; s.out = h->std_sums;
      TRY(launch(h, "std_sums", k_std_sums, dim3(1), dim3(kThreads), 0, s));
    }
  }
  if (phase == 1) return DSACT_OK;
  // pipelined graph, update whose policy backward rides in the next forward launch: critics' backward + their tiles + close
  if (h->bqt_now && h->pipe_defer_now && actor_backward && phase == 2) return enqueue_chain_bwd_qt(h, fused, ride);
  // ... update that moves the policy: the whole backward as one launch
  if (h->bqp_now && !h->pipe_defer_now && actor_backward && phase == 2) return enqueue_chain_bwd_qpt(h, fused, ride);
  TRY(enqueue_chain_bwd_q(h, (actor_backward ? 2 : 1) * h->nq, ride));
  // CNN nets (batch <= 1024: one gradient arena): dL/d features = dZ0 . W0[:, :F] right behind the chains that produce dZ0 and
  // before the launch whose tiles update W0; the conv stacks' backward follows the MLP part (enqueue_grads' order)
  // (full update on one gradient arena: ONE dfeat launch for the three nets behind the policy chain, see below)
  if (h->cnn && !one_dfeat) TRY(run_stage(h, h->dfeat_q));
  if (!actor_backward) {
    if (h->dw_chunks == 1) {
      TRY(run_dw2(h, off[0], off[2], fused, fused));
      if (h->cnn) TRY(enqueue_conv_backward(h, h->nq, fused));
      return DSACT_OK;
    }
    TRY(run_dw2(h, off[0], off[2], false, false));
    if (fused) return enqueue_adam(h, true);
    return sum_parts(h, 0, (size_t)h->nq * h->n_q);
  }
  if (phase == 3) {
    TRY(run_dw2(h, off[0], off[2], false, false));
    if (h->cnn) TRY(enqueue_conv_backward(h, h->nq, false));
    return sum_parts(h, 0, (size_t)h->nq * h->n_q);
  }
actor_part:
  if (phase == 4) {
    TRY(enqueue_chain_bwd_pi(h, 0, 0, false));
    if (h->cnn) TRY(run_stage(h, h->dfeat_pi));
    TRY(run_dw2(h, h->dw2_off[2], h->dw2_off[3], false, false));
    if (h->cnn) TRY(enqueue_conv_backward(h, 1, false, h->nq));
    return sum_parts(h, (size_t)h->nq * h->n_q, h->n_online - 1);
  }
  // the critics' dW (+ Adam) tiles ride in the policy-backward launch on the CUs its 32 chain workgroups leave idle
  {
    int ride_end = h->dw2_off[2];
    // batch <= 256: the policy's own tiles and the closing block ride in the same launch behind the riders and wait for
    // the chain's arrival counter (k_chain_bwd_pi, merge_dw) -- one launch and one kernel boundary less per update
    if (h->pi_merge && h->dw_chunks == 1 && ride_end == h->dw2_off[2]) {
      if (h->pipe_defer_now) return enqueue_chain_bwd_pi_close(h, fused);   // the policy backward rides in the next forward launch
      return enqueue_chain_bwd_pi(h, h->dw2_off[0], ride_end, fused, true);
    }
    if (one_dfeat) {
      // riders: the critics' hidden / output layer tiles; their first-layer tiles (Adam on W0) follow the dfeat launch
      TRY(enqueue_chain_bwd_pi(h, h->dw2_off[0], h->dw2_mid, fused));
      TRY(run_stage(h, h->dfeat_all));
      TRY(run_dw2(h, h->dw2_mid, h->dw2_off[3], fused, fused));
      return enqueue_conv_backward(h, h->nq + 1, fused);
    }
    TRY(enqueue_chain_bwd_pi(h, h->dw2_off[0], ride_end, fused));
    if (h->cnn) TRY(run_stage(h, h->dfeat_pi));   // needs the policy's W0 BEFORE the fused Adam of the next launch
    if (h->dw_chunks == 1) {
      TRY(run_dw2(h, ride_end, h->dw2_off[3], fused, fused));
      if (h->cnn) TRY(enqueue_conv_backward(h, h->nq + 1, fused));
      return DSACT_OK;
    }
    TRY(run_dw2(h, ride_end, h->dw2_off[3], false, false));
  }
  if (fused) return enqueue_adam(h, true);
  return sum_parts(h, 0, h->n_online - 1);
}

// phase 0: everything; 1: forward part up to the local {sum std1, sum std2} (strict data-parallel mode: the
// caller all-reduces those two floats); 2: loss + backward (+ fused update);
// 3 / 4 (data-parallel overlap, unfused): 3 = everything up to and including the critics' gradients (q1 | q2
// segment of the arena final), 4 = actor part (policy | log_alpha | mean_std tail) -- the caller starts the
// all-reduce of the first segment between the two
// `ride` (graph replays with the merged gather): nullptr, or the riders of the loss launch -- this update's
// bookkeeping and, when ride->n_gather > 0, the next update's gather into the other batch set
int enqueue_grads(dsact_handle* h, bool actor_backward, bool fused, int phase = 0, const RideArgs* ride = nullptr) {
  h->have_local_tail = true;
  h->uev_valid = false;   // dsact_step / dsact_compute_grads set it again once their closing event is recorded
  if (h->chain_ok) return enqueue_grads_chain(h, actor_backward, fused, phase, ride);
  const int B = h->B, A = h->A;
  const int Lq = h->Lq, Lp = h->Lp;   // hidden layers of the critics / the policy nets (equal unless policy_n_hidden is set)
  if (phase == 4) goto actor_part;
  if (phase != 2) {
  if (h->cnn) TRY(enqueue_conv_forward(h));
  for (size_t l = 0; l < h->fwd1.size(); ++l) TRY(run_stage(h, h->fwd1[l]));
  {
    HeadsArgs a;
    memset(&a, 0, sizeof(a));
    const int chs[4] = {C_PI, C_PIT, C_Q1C, C_Q2C};
    const int n_heads = h->nq == 2 ? 4 : 3;
    a.v1_stats = h->nq == 1;
    for (int i = 0; i < 4; ++i) {
      const int net = kChainNet[chs[i]];
      const NetDesc& d = net_desc(h, net);
      const int L = chain_L(h, chs[i]);
      a.H[i] = h->Hb[chs[i]][L - 1];
      a.Wout[i] = net_params(h, net) + d.w_off[L];
      a.bout[i] = net_params(h, net) + d.b_off[L];
    }
    a.Wch[0] = a.Wch[1] = h->wp[Lp - 1]; a.Wch[2] = a.Wch[3] = h->wq[Lq - 1];
    a.W = a.Wch[0] > a.Wch[2] ? a.Wch[0] : a.Wch[2];
    a.B = B; a.O = h->F; a.A = A; a.ldx = h->ldx;
    a.eps_new = h->eps_new; a.eps_2 = h->eps_2;
    a.XP = h->Xc[C_Q1P]; a.XPb = h->Xc[C_Q2P] != h->Xc[C_Q1P] ? h->Xc[C_Q2P] : nullptr;
    a.X2 = h->Xc[C_Q1T]; a.X2b = h->Xc[C_Q2T] != h->Xc[C_Q1T] ? h->Xc[C_Q2T] : nullptr;
    a.logits_pi = h->logits_pi; a.logits_pit = h->logits_pit; a.logp_new = h->logp_new; a.logp2 = h->logp2;
    a.qout[0] = h->qout_c[0]; a.qout[1] = h->qout_c[1];
    a.qstd[0] = h->qstd_c[0]; a.qstd[1] = h->qstd_c[1];
    a.part_heads = h->part_heads; a.act_scale = h->act_scale; a.act_center = h->act_center;
    a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std;
    a.q_out_act = h->cfg.value_out_act; a.pi_out_act = h->cfg.policy_out_act; a.pi_out_n = h->cfg.policy_std_param ? A : 2 * A;
    a.qdmean[0] = h->qdmean[0]; a.qdmean[1] = h->qdmean[1]; a.pi_dact = h->pi_dact;
    a.timeline = tl_for(h, "heads");
#define CALL_HEADS(N) TRY(launch(h, "heads", k_heads<N>, dim3(h->n_heads_wg, n_heads), dim3(kThreads), 0, a))
    NCH_DISPATCH(a.W, CALL_HEADS);
  }
  for (size_t l = 0; l < h->fwd2.size(); ++l) TRY(run_stage(h, h->fwd2[l]));
  if (h->use_std_sums || h->auto_std_sums) {
    // large batches (and the strict data-parallel mode, where the caller all-reduces std_sums
    // between the phases): the std column is summed by one workgroup instead of by every wave
    StdSumArgs s;
    s.qstd_c[0] = h->qstd_c[0]; s.qstd_c[1] = h->qstd_c[1]; s.B = B; s.out = h->std_sums;
    TRY(launch(h, "std_sums", k_std_sums, dim3(1), dim3(kThreads), 0, s));
  }
  }  // phase != 2
  if (phase == 1) return DSACT_OK;
  {
  const int L = Lq;   // the loss kernels read the critics' last hidden layer
  if (h->nq == 1) {
    LossV1Args a;
    memset(&a, 0, sizeof(a));
    const int chs[2] = {C_Q1T, C_Q1P};
    for (int i = 0; i < 2; ++i) {
      const int net = kChainNet[chs[i]];
      a.Hl[i] = h->Hb[chs[i]][L - 1];
      a.Wout[i] = net_params(h, net) + h->qd.w_off[L];
      a.bout[i] = net_params(h, net) + h->qd.b_off[L];
    }
    const int dch[2] = {C_Q1C, C_Q1P};
    for (int i = 0; i < 2; ++i) {
      a.Gl[i] = h->Gb[dch[i]][L - 1];
      a.dZl[i] = h->dZ[kDzSlot[dch[i]]][L - 1];
    }
    a.dout[0] = h->dout[0]; a.dout[1] = h->dout[2];   // same slots as DSAC_V2's q1c / q1p
    a.qout_c = h->qout_c[0]; a.qstd_c = h->qstd_c[0]; a.qout_t = h->qout_t[0]; a.qout_p = h->qout_p[0];
    a.rew = h->rew; a.done = h->done; a.logp2 = h->logp2; a.logp_new = h->logp_new; a.z_t = h->z5;
    a.log_alpha = h->online + h->n_online - 1;
    a.part_loss = h->part_loss; a.grads_tail = h->grads + h->n_online;
    a.W = h->wq[L - 1]; a.B = B; a.inv_B = 1.0f / (float)B;
    a.auto_alpha = h->cfg.auto_alpha; a.alpha_fixed = h->cfg.alpha_fixed; a.gamma = h->cfg.gamma; a.td_bound = h->cfg.td_bound; a.bound = h->cfg.v1_unbounded ? 0 : 1;
    if (ride) a.ride = *ride;
    a.ride.n_loss_blocks = h->n_loss_wg;
    const int n_riders = ride ? ride->n_gather + (ride->bookkeeping ? 1 : 0) : 0;
#define CALL_LOSS1(N) TRY(launch(h, "loss", k_loss_v1<N>, dim3(h->n_loss_wg + n_riders), dim3(kThreads), 0, a))
    NCH_DISPATCH(a.W, CALL_LOSS1);
  } else {
    LossArgs a;
    memset(&a, 0, sizeof(a));
    const int chs[4] = {C_Q1T, C_Q2T, C_Q1P, C_Q2P};
    for (int i = 0; i < 4; ++i) {
      const int net = kChainNet[chs[i]];
      a.Hl[i] = h->Hb[chs[i]][L - 1];
      a.Wout[i] = net_params(h, net) + h->qd.w_off[L];
      a.bout[i] = net_params(h, net) + h->qd.b_off[L];
    }
    const int dch[4] = {C_Q1C, C_Q2C, C_Q1P, C_Q2P};
    for (int i = 0; i < 4; ++i) {
      a.Gl[i] = h->Gb[dch[i]][L - 1];
      a.dZl[i] = h->dZ[kDzSlot[dch[i]]][L - 1];
      a.dout[i] = h->dout[i];
    }
    for (int i = 0; i < 2; ++i) { a.qout_c[i] = h->qout_c[i]; a.qstd_c[i] = h->qstd_c[i]; a.qout_t[i] = h->qout_t[i]; a.qout_p[i] = h->qout_p[i]; }
    a.rew = h->rew; a.done = h->done; a.logp2 = h->logp2; a.logp_new = h->logp_new; a.z5 = h->z5; a.z6 = h->z6;
    a.log_alpha = h->online + h->n_online - 1;
    a.part_loss = h->part_loss; a.grads_tail = h->grads + h->n_online; a.st = h->st;
    a.W = h->wq[L - 1]; a.B = B;
    a.inv_B = 1.0f / (float)B;
    a.inv_Bg = h->use_std_sums ? 1.0f / (float)h->cfg.global_batch : 1.0f / (float)B;
    a.std_sums = (h->use_std_sums || h->auto_std_sums) ? h->std_sums : nullptr;
    a.auto_alpha = h->cfg.auto_alpha; a.alpha_fixed = h->cfg.alpha_fixed; a.gamma = h->cfg.gamma; a.tau_b = h->cfg.tau_b; a.one_minus_tau_b = (float)(1.0 - h->cfg.tau_b);
    a.q_out_act = h->cfg.value_out_act;
    a.qdmean[0] = h->qdmean[0]; a.qdmean[1] = h->qdmean[1];
    a.timeline = tl_for(h, "loss");
    if (ride) a.ride = *ride;
    a.ride.n_loss_blocks = h->n_loss_wg;
    const int n_riders = ride ? ride->n_gather + (ride->bookkeeping ? 1 : 0) : 0;
#define CALL_LOSS(N) TRY(launch(h, "loss", k_loss<N>, dim3(h->n_loss_wg + n_riders), dim3(kThreads), 0, a))
    NCH_DISPATCH(a.W, CALL_LOSS);
  }
  }
  if (!actor_backward) {
    // off iteration of the delayed update: the reference computes the actor / alpha gradients and
    // discards them (dsac_v2.py:174-186 vs :324) -- only the critics' backward is needed
    for (size_t i = 0; i < h->bwdq_critic.size(); ++i) TRY(run_stage(h, h->bwdq_critic[i]));
    if (h->cnn) TRY(run_stage(h, h->dfeat_q));
    if (h->dw_chunks == 1) {
      TRY(run_dw(h, h->dw_off[0], h->dw_off[2], fused, fused));
      if (h->cnn) TRY(enqueue_conv_backward(h, h->nq, fused));
    } else {
      TRY(run_dw(h, h->dw_off[0], h->dw_off[2], false, false));
      if (h->cnn) TRY(enqueue_conv_backward(h, h->nq, false));
      if (fused) TRY(enqueue_adam(h, true));   // policy segment: not updated on off iterations, partials ignored
      else TRY(sum_parts(h, 0, (size_t)h->nq * h->n_q));
    }
    return DSACT_OK;
  }
  for (size_t i = 0; i < h->bwdq.size(); ++i) TRY(run_stage(h, h->bwdq[i]));
  if (h->cnn) TRY(run_stage(h, h->dfeat_q));   // reads the step's padded copy of W0: safe against the fused Adam below
  if (phase == 3) {
    TRY(run_dw(h, h->dw_off[0], h->dw_off[2], false, false));
    if (h->cnn) TRY(enqueue_conv_backward(h, h->nq, false));
    TRY(sum_parts(h, 0, (size_t)h->nq * h->n_q));
    return DSACT_OK;
  }
actor_part:
  // The critics' weight-gradient tiles ride along in the under-filled launches of the actor chain: heads_bwd (B/4 row
  // blocks) and the policy-backward stages (one problem each). Spread evenly so that each launch stays within one
  // round of workgroups (64 + 177 <= 256 CUs at batch 256; two carriers made it 64 + 265).
  const int n_carriers = (int)h->bwdpi.size() + 1;
  const int crit_tiles = h->dw_off[2] - h->dw_off[0];
  // tiles that may ride: the critics' + (behind heads_bwd only) the policy's output layer
  const int ride_end = h->bwdpi.empty() ? h->dw_off[2] : h->dw_pol_rest;
  int ride_hb = 0;
  if (phase == 0 && !(h->use_fork && !h->profiling)) {
    ride_hb = (ride_end - h->dw_off[0]) / n_carriers;
    if (ride_hb > crit_tiles) ride_hb = crit_tiles;
  }
  if (h->use_fork && !h->profiling) {
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
    TRY(run_dw(h, h->dw_off[0], h->dw_off[2], fused, false, h->aux_stream));
    HIPCHK(h, hipEventRecord(h->ev_join, h->aux_stream));
  }
  {
    HeadsBwdArgs a;
    a.dZ1[0] = h->dZ[kDzSlot[C_Q1P]][0]; a.dZ1[1] = h->dZ[kDzSlot[C_Q2P]][0];
    a.W1aT[0] = h->W1aT[0]; a.W1aT[1] = h->W1aT[1]; a.W0 = h->wq[0];
    a.logits_pi = h->logits_pi; a.eps_new = h->eps_new; a.log_alpha = h->online + h->n_online - 1;
    a.Wout_pi = net_params(h, N_POL) + h->pd.w_off[Lp];
    a.G_pi = h->Gb[C_PI][Lp - 1]; a.dZ_pi = h->dZ[kDzSlot[C_PI]][Lp - 1];
    a.dout_pi = h->dout_pi; a.d_new_act = h->d_new_act;
    a.WL = h->wp[Lp - 1]; a.B = B; a.O = h->F; a.A = A;
    a.inv_B = 1.0f / (float)B; a.auto_alpha = h->cfg.auto_alpha; a.alpha_fixed = h->cfg.alpha_fixed;
    a.act_scale = h->act_scale; a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std;
    a.part_loss = h->part_loss; a.n_part = B; a.target_entropy = -(float)A;
    a.grad_log_alpha = h->grads + h->n_online - 1;
    a.pi_out_act = h->cfg.policy_out_act; a.pi_out_n = h->cfg.policy_std_param ? A : 2 * A;
    a.pi_dact = h->pi_dact;
    a.timeline = tl_for(h, "heads_bwd");
    a.n_row_blocks = (B + 3) / 4;
    a.extra = h->d_tiles; a.n_extra = 0;
    a.fo = fused_opt(h, fused);
    if (ride_hb > 0) { a.extra = h->d_tiles + h->dw_off[0]; a.n_extra = ride_hb; }
    const size_t hb_lds = a.n_extra ? tile_lds_bytes(dw_k(h)) : 0;
#define CALL_HBWD(N) TRY(launch(h, "heads_bwd", k_heads_bwd<N>, dim3(a.n_row_blocks + a.n_extra), dim3(kThreads), hb_lds, a))
    NCH_DISPATCH(a.WL, CALL_HBWD);
  }
  const size_t np = h->bwdpi.size();
  if (h->use_fork && !h->profiling) {
    // The critics' weight gradients (+ their Adam/Polyak) depend only on the critic backward above,
    // not on the actor chain (heads_bwd -> policy backward): they run on a forked branch -- a second
    // HIP stream, captured as a parallel branch of the graph -- and join before the final launch.
    // (heads_bwd was already enqueued on the main stream; the fork event is recorded before it.)
    for (size_t i = 0; i < np; ++i) TRY(run_stage(h, h->bwdpi[i], 0, 0, fused));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    TRY(run_dw(h, h->dw_off[2], h->dw_off[3], fused, fused));
    return DSACT_OK;
  }
  if (phase == 4) {
    for (size_t i = 0; i < np; ++i) TRY(run_stage(h, h->bwdpi[i]));
    if (h->cnn) TRY(run_stage(h, h->dfeat_pi));
    TRY(run_dw(h, h->dw_off[2], h->dw_off[3], false, false));
    if (h->cnn) TRY(enqueue_conv_backward(h, 1, false, h->nq));
    TRY(sum_parts(h, (size_t)h->nq * h->n_q, h->n_online - 1));
    return DSACT_OK;
  }
  // unforked: the rest of the critics' tiles ride along in the policy-backward launches, evenly
  {
    int x = h->dw_off[0] + ride_hb;
    for (size_t i = 0; i < np; ++i) {
      const int left = ride_end - x;
      const int take = i + 1 == np ? left : left / (int)(np - i);
      TRY(run_stage(h, h->bwdpi[i], x, x + take, fused));
      x += take;
    }
    if (np == 0 && ride_hb) { /* no policy-backward stage: the final launch takes what heads_bwd did not */ }
  }
  if (h->cnn) TRY(run_stage(h, h->dfeat_pi));  // needs the policy's W0 BEFORE the fused Adam of the next launch
  // policy weight gradients (+ the critics' when there was no launch to ride in) + close of the update
  if (h->dw_chunks == 1) {
    TRY(run_dw(h, np ? ride_end : h->dw_off[0] + ride_hb, h->dw_off[3], fused, fused));
    if (h->cnn) TRY(enqueue_conv_backward(h, h->nq + 1, fused));
  } else {
    TRY(run_dw(h, np ? ride_end : h->dw_off[0] + ride_hb, h->dw_off[3], false, false));
    if (h->cnn) TRY(enqueue_conv_backward(h, h->nq + 1, false));
    if (fused) TRY(enqueue_adam(h, true));
    else TRY(sum_parts(h, 0, h->n_online - 1));
  }
  return DSACT_OK;
}

// from_parts: split-K flow of the fused step -- the optimiser sums the chunk partials itself (no k_sum_parts pass)
int enqueue_adam(dsact_handle* h, bool from_parts) {
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  if (from_parts && h->dw_chunks > 1) {
    a.part = h->dw_parts; a.part_stride = (long long)h->dw_part_stride; a.n_part = h->dw_chunks;
    const int nets[3] = {N_Q1, N_Q2, N_POL};
    for (int i = 0; i < 3; ++i) {   // conv parameters (CNN nets) are reduced by their own kernel straight into `grads`
      const long long b = (long long)(net_grads(h, nets[i]) - h->grads);
      a.direct_lo[i] = b; a.direct_hi[i] = b + (long long)net_desc(h, nets[i]).w_off[0];
    }
  }
  a.p = h->online; a.tgt = h->target; a.m = h->adam_m; a.v = h->adam_v; a.g = h->grads;
  a.n_q2 = (long long)(h->nq * h->n_q); a.n_online3 = (long long)(h->nq * h->n_q + h->n_pi); a.n_total = (long long)h->n_online;
  a.st = h->st;
  a.b1w = (float)(1.0 - h->cfg.adam_beta1);
  a.beta2 = (float)h->cfg.adam_beta2;
  a.b2w = (float)(1.0 - h->cfg.adam_beta2);
  a.eps = h->cfg.adam_eps;
  const double polyak = 1.0 - h->cfg.tau;  // dsac_v2.py:331
  a.polyak = (float)polyak;
  a.one_minus_polyak = (float)(1.0 - polyak);
  a.auto_alpha = h->cfg.auto_alpha;
  a.commit_ms = h->have_local_tail ? 1 : 0;   // dsac_v2.py remote_update never touches mean_std: commit only a tail this handle computed
  h->have_local_tail = false;
  const int blocks = (int)((h->n_online + kThreads * 8 - 1) / (kThreads * 8));  // 2 float4 groups per thread
  return launch(h, "adam_polyak", k_adam, dim3(blocks), dim3(kThreads), 0, a);
}

static void destroy_graph_set(dsact_handle::GraphSet& g) {
  if (g.exec) hipGraphExecDestroy(g.exec);
  if (g.graph) hipGraphDestroy(g.graph);
  for (int p = 0; p < dsact_handle::kPipePhases; ++p) {
    if (g.pexec[p]) hipGraphExecDestroy(g.pexec[p]);
    if (g.pgraph[p]) hipGraphDestroy(g.pgraph[p]);
    if (g.pargs[p]) hipFree(g.pargs[p]);
  }
  g = dsact_handle::GraphSet();
}
// the active graph (if any) moves into the cache; nothing is destroyed
static void stash_graph(dsact_handle* h) {
  if (!(h->graph_exec != nullptr || h->pipe_graph)) return;
  dsact_handle::GraphSet g;
  g.graph = h->graph; g.exec = h->graph_exec;
  for (int p = 0; p < dsact_handle::kPipePhases; ++p) { g.pgraph[p] = h->pgraph[p]; g.pexec[p] = h->pexec[p]; g.pargs[p] = h->pargs[p]; }
  g.pipe = h->pipe_graph; g.merged = h->merged_graph; g.noise_table = h->graph_noise_table;
  g.steps = h->graph_steps; g.flags = h->graph_flags;
  h->graph_cache.push_back(g);
  h->graph = nullptr; h->graph_exec = nullptr;
  for (int p = 0; p < dsact_handle::kPipePhases; ++p) { h->pgraph[p] = nullptr; h->pexec[p] = nullptr; h->pargs[p] = nullptr; }
  h->pipe_graph = false; h->graph_steps = 0; h->graph_noise_table = false;
}
// a cached graph of that shape becomes the active one (the caller stashed the previous one); false: none cached
static bool activate_graph(dsact_handle* h, int steps, uint32_t flags, bool noise_table) {
  for (size_t i = 0; i < h->graph_cache.size(); ++i) {
    dsact_handle::GraphSet& g = h->graph_cache[i];
    if (g.steps != steps || g.flags != flags || g.noise_table != noise_table) continue;
    h->graph = g.graph; h->graph_exec = g.exec;
    for (int p = 0; p < dsact_handle::kPipePhases; ++p) { h->pgraph[p] = g.pgraph[p]; h->pexec[p] = g.pexec[p]; h->pargs[p] = (PipeFwd*)g.pargs[p]; }
    h->pipe_graph = g.pipe; h->merged_graph = g.merged; h->graph_noise_table = g.noise_table;
    h->graph_steps = g.steps; h->graph_flags = g.flags;
    h->graph_cache.erase(h->graph_cache.begin() + (long)i);
    return true;
  }
  return false;
}
// every captured graph goes: the active one and the cached ones (they bake in pointers, hyper-parameters, launch forms)
// the ACTIVE graph only (a failed or abandoned capture): cached graphs of other shapes stay valid
static void drop_active_graph(dsact_handle* h) {
  if (h->graph_exec) { hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
  if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
  for (int p = 0; p < dsact_handle::kPipePhases; ++p) {
    if (h->pexec[p]) { hipGraphExecDestroy(h->pexec[p]); h->pexec[p] = nullptr; }
    if (h->pgraph[p]) { hipGraphDestroy(h->pgraph[p]); h->pgraph[p] = nullptr; }
    if (h->pargs[p]) { hipFree(h->pargs[p]); h->pargs[p] = nullptr; }
  }
  h->pipe_graph = false;
  h->graph_steps = 0;
  h->graph_noise_table = false;
}
void drop_graphs(dsact_handle* h) {
  if (h->graph_exec) { hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
  if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
  for (int p = 0; p < dsact_handle::kPipePhases; ++p) {
    if (h->pexec[p]) { hipGraphExecDestroy(h->pexec[p]); h->pexec[p] = nullptr; }
    if (h->pgraph[p]) { hipGraphDestroy(h->pgraph[p]); h->pgraph[p] = nullptr; }
    if (h->pargs[p]) { hipFree(h->pargs[p]); h->pargs[p] = nullptr; }
  }
  for (auto& g : h->graph_cache) destroy_graph_set(g);
  h->graph_cache.clear();
  h->pipe_graph = false;
  h->graph_steps = 0;
  h->graph_noise_table = false;
}
bool have_graph(const dsact_handle* h) { return h->graph_exec != nullptr || h->pipe_graph; }
// active OR cached: what the "baked into the captured graph" refusals have to look at (dsact_run_group keeps graphs of other
// group lengths in the cache, and a failed build may leave no active graph beside them)
bool any_graph(const dsact_handle* h) { return have_graph(h) || !h->graph_cache.empty(); }

// In-launch hand-overs (merged forward / backward launches) use BOUNDED spins: a consumer that waited ~0.1 s for its
// producers' flags gives up, computes on whatever it finds and writes the hand-off word in mapped host memory. Every
// entry point comes through here: the call fails (DSACT_E_HIP), the merged launches are switched off for this handle
// (the plain multi-launch chain has no in-launch dependencies) and a captured graph is captured again without them, so
// the NEXT call runs -- on device state the caller must treat as invalid (restore a checkpoint / re-bind the arenas).
int check_handoff(dsact_handle* h) {
  // two words in mapped host memory (one overwritten word let an acting-forward timeout hide an update kernel's):
  // [0] is raised by the update kernels, [1] by the acting forward (k_act_mlp) -- neither store can erase the other
  volatile int* const w_upd = (volatile int*)h->handoff_host;
  volatile int* const w_act = w_upd ? w_upd + 1 : nullptr;
  if (!h->handoff_host || h->in_handoff || !(*w_upd | *w_act)) return DSACT_OK;
  h->in_handoff = true;
  const hipError_t e_dev = hipSetDevice(h->device);
  const hipError_t e_sync = hipStreamSynchronize(h->stream);
  const int upd = *w_upd;   // (read after the drain: an update kernel may have given up too while the stream emptied)
  *w_upd = 0; *w_act = 0;
  if (!upd) {
    // raised by the acting forward only: it reads the parameters and writes nothing the update path reads, so the training
    // state is intact -- the call fails, nothing is switched off, nothing has to be restored
    h->in_handoff = false;
    h->handoff_failures += 1;
    return fail(h, DSACT_E_HIP, "the acting forward's layer hand-over timed out (a wave waited > 0.1 s for its input): this "
                                "call's action is invalid; parameters and optimiser state are untouched%s%s",
                e_sync != hipSuccess ? "; HIP also reported: " : "", e_sync != hipSuccess ? hipGetErrorString(e_sync) : "");
  }
  const bool had_graph = have_graph(h);
  const int steps = had_graph ? h->graph_steps : h->want_graph_steps;
  const uint32_t gflags = had_graph ? h->graph_flags : h->want_graph_flags;
  const bool noise_tab = h->graph_noise_table;
  h->fwd_merge = false;
  h->pi_merge = false;
  h->handoff_failures += 1;
  drop_graphs(h);
  int twin_rc = DSACT_OK;
  if (h->twin_par) {   // CNN nets: both trunks of a net in one workgroup, groups A and B as two launches -- no in-launch waits left
    h->env_twin_seq = true;
    h->debug_withhold = 0;
    twin_rc = build_twin_fwd(h);
  }
  hipError_t e_set = hipSuccess;
  if (h->chain_flags) e_set = hipMemset(h->chain_flags, 0, kChainFlagInts * sizeof(int));
  h->flags_dirty = false;
  int rebuilt = DSACT_E_STATE;
  const bool wanted = had_graph || steps > 0;
  if (wanted && twin_rc == DSACT_OK) {
    // (the re-capture must not trip over an earlier, still unacknowledged timeout: capturing launches nothing)
    h->state_invalid = false;
    h->noise_table_on = noise_tab;
    rebuilt = dsact_graph_build(h, steps, gflags);
    h->noise_table_on = false;
    if (rebuilt == DSACT_OK) h->graph_noise_table = noise_tab;
  }
  // not re-captured: remembered, and built by the first replay after the caller's acknowledgement
  h->want_graph_steps = (wanted && rebuilt != DSACT_OK) ? steps : 0;
  h->want_graph_flags = gflags;
  h->in_handoff = false;
  h->state_invalid = true;   // (after the re-capture: dsact_graph_build itself is an entry point that refuses an invalid state)
  const hipError_t e_hip = e_dev != hipSuccess ? e_dev : e_sync != hipSuccess ? e_sync : e_set;
  return fail(h, DSACT_E_HIP,
              "an in-launch hand-over timed out: a workgroup waited > 0.1 s for its producers' ready flags, so every result "
              "since the last successful call is invalid -- parameters, optimiser moments and targets included: restore them, "
              "then acknowledge with dsact_set_state(adam_steps, mean_std), dsact_debug_set(\"ack_state\") or dsact_bind_arenas "
              "(update entry points fail until then). Merged launches are now disabled for this handle%s%s%s",
              wanted ? (rebuilt == DSACT_OK ? "; the graph was captured again without them" : "; re-capturing the graph failed (it is built again by the next replay)")
                     : "",
              e_hip != hipSuccess ? "; HIP also reported: " : "", e_hip != hipSuccess ? hipGetErrorString(e_hip) : "");
}

// a graph a hand-over failure could not re-capture is captured by the first replay after the acknowledgement
static int ensure_wanted_graph(dsact_handle* h) {
  if (have_graph(h) || h->want_graph_steps <= 0) return DSACT_OK;
  const int steps = h->want_graph_steps;
  h->want_graph_steps = 0;
  return dsact_graph_build(h, steps, h->want_graph_flags);
}

int check_ready(dsact_handle* h, bool need_batch) {
  if (!h) return DSACT_E_INVALID;
  TRY(check_handoff(h));
  if (h->state_invalid)
    return fail(h, DSACT_E_STATE, "device state is invalid after a hand-over timeout: restore parameters / optimiser state, then "
                                  "call dsact_set_state or dsact_bind_arenas");
  if (!h->online || !h->grads) return fail(h, DSACT_E_STATE, "arenas not bound (dsact_bind_arenas)");
  if (!h->limits_set) return fail(h, DSACT_E_STATE, "action limits not set (dsact_set_action_limits)");
  if (need_batch && !h->have_batch) return fail(h, DSACT_E_STATE, "no minibatch staged (dsact_gather / dsact_load_batch)");
  h->dev_it_next = -1;   // every update entry point except the graph replays passes through here
  return DSACT_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int dsact_version(void) { return 1; }

int dsact_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dsact_last_error(const dsact_handle* h) { return h ? h->err : "null handle"; }

int dsact_create(const dsact_config* cfg, int device, dsact_handle** out) {
  if (!cfg || !out) return DSACT_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return DSACT_E_NODEVICE;
  if (device < 0 || device >= ndev) return DSACT_E_INVALID;
  dsact_handle* h = new dsact_handle();
  h->cfg = *cfg;
  h->device = device;
  *out = h;  // returned even on failure so that the caller can read the message, then destroy
  if (cfg->obs_dim < 1 || cfg->act_dim < 1 || cfg->act_dim > 32) return fail(h, DSACT_E_INVALID, "act_dim must be in 1..32 (got %d), obs_dim >= 1", cfg->act_dim);
  if (cfg->n_hidden < 1 || cfg->n_hidden > DSACT_MAX_HIDDEN_LAYERS) return fail(h, DSACT_E_INVALID, "n_hidden must be 1..%d", DSACT_MAX_HIDDEN_LAYERS);
  for (int l = 0; l < cfg->n_hidden; ++l)
    if (cfg->hidden[l] < 1 || cfg->hidden[l] > kMaxWidth) return fail(h, DSACT_E_INVALID, "hidden width must be 1..%d", kMaxWidth);
  if (cfg->batch < 1) return fail(h, DSACT_E_INVALID, "batch must be >= 1");
  if (cfg->delay_update < 1) return fail(h, DSACT_E_INVALID, "delay_update must be >= 1");
  if (cfg->value_act < 0 || cfg->value_act > ACT_TANH || cfg->policy_act < 0 || cfg->policy_act > ACT_TANH)
    return fail(h, DSACT_E_INVALID, "hidden activation must be 0..5 (gelu, relu, elu, selu, sigmoid, tanh)");
  if (cfg->act_dist != 0 && cfg->act_dist != 1) return fail(h, DSACT_E_INVALID, "act_dist must be 0 (TanhGaussDistribution) or 1 (GaussDistribution)");
  if (cfg->policy_std_param != 0 && cfg->policy_std_param != 1) return fail(h, DSACT_E_INVALID, "policy_std_param must be 0 (mlp_shared) or 1 (parameter)");
  if (cfg->policy_std_param && (cfg->conv_type != DSACT_CONV_NONE || cfg->algo != 0))
    return fail(h, DSACT_E_INVALID, "policy_std_type 'parameter' is built for DSAC_V2 with MLP nets");
  if (cfg->policy_twin != 0 && cfg->policy_twin != 1) return fail(h, DSACT_E_INVALID, "policy_twin must be 0 or 1 (policy_std_type mlp_separated)");
  if (cfg->policy_twin && (cfg->conv_type != DSACT_CONV_NONE || cfg->algo != 0 || cfg->policy_std_param))
    return fail(h, DSACT_E_INVALID, "policy_std_type 'mlp_separated' is built for DSAC_V2 with MLP nets (and excludes 'parameter')");
  for (int oa : {cfg->value_out_act, cfg->policy_out_act})
    if (oa != 0 && (oa < ACT_RELU || oa > ACT_TANH) && oa != OUT_ACT_GELU)
      return fail(h, DSACT_E_INVALID, "output activation must be 0 (linear), 1..5 (relu, elu, selu, sigmoid, tanh) or 6 (gelu)");
  if ((cfg->value_out_act || cfg->policy_out_act) && (cfg->conv_type != DSACT_CONV_NONE || cfg->algo != 0))
    return fail(h, DSACT_E_INVALID, "output activations other than linear are built for DSAC_V2 with MLP nets (tile-stage kernels)");
  if (h->cfg.global_batch < h->cfg.batch) h->cfg.global_batch = h->cfg.batch;
  HIPCHK(h, hipSetDevice(device));
  h->O = cfg->obs_dim; h->A = cfg->act_dim; h->L = cfg->n_hidden; h->B = cfg->batch;   // (L: the deeper family once policy_n_hidden is read)
  h->F = h->O;
  h->Brows = h->B > kActRows ? h->B : kActRows;
  int nblk = 1;
  if (cfg->conv_type != DSACT_CONV_NONE) {
    // networks/cnn.py:173-228: the two conv stacks the reference defines
    static const int k1[] = {8, 4, 3}, c1[] = {32, 64, 64}, s1[] = {4, 2, 1};
    static const int k2[] = {4, 3, 3, 3, 3, 3}, c2[] = {8, 16, 32, 64, 128, 256}, s2[] = {2, 2, 2, 2, 1, 1};
    if (cfg->conv_type != DSACT_CONV_TYPE_1 && cfg->conv_type != DSACT_CONV_TYPE_2) return fail(h, DSACT_E_INVALID, "conv_type must be 0, 1 (type_1) or 2 (type_2)");
    const bool t1 = cfg->conv_type == DSACT_CONV_TYPE_1;
    const int* ks = t1 ? k1 : k2; const int* cs = t1 ? c1 : c2; const int* ss = t1 ? s1 : s2;
    h->n_conv = t1 ? 3 : 6;
    if (cfg->img_c < 1 || cfg->img_h < 1 || cfg->img_w < 1 || (long long)cfg->img_c * cfg->img_h * cfg->img_w != cfg->obs_dim)
      return fail(h, DSACT_E_INVALID, "obs_dim must equal img_c*img_h*img_w for CNN nets");
    if (cfg->obs_dim % 4) return fail(h, DSACT_E_INVALID, "img_c*img_h*img_w must be a multiple of 4");
    int C = cfg->img_c, H = cfg->img_h, W = cfg->img_w;
    for (int j = 0; j < h->n_conv; ++j) {
      ConvGeom& g = h->cg[j];
      g.H = H; g.W = W; g.Cin = C; g.KS = ks[j]; g.stride = ss[j]; g.Cout = cs[j];
      g.OH = (H - g.KS) / g.stride + 1; g.OW = (W - g.KS) / g.stride + 1;
      if (H < g.KS || W < g.KS || g.OH < 1 || g.OW < 1) return fail(h, DSACT_E_INVALID, "image %dx%d too small for conv layer %d", cfg->img_h, cfg->img_w, j);
      g.K = g.KS * g.KS * g.Cin; g.KWC = g.KS * g.Cin; g.rowskip = g.W * g.Cin - g.KWC;
      g.inv_kwc = 1.0f / (float)g.KWC;
      if (g.KWC % 4) return fail(h, DSACT_E_INVALID, "kernel_width*channels of conv layer %d must be a multiple of 4 (got %d)", j, g.KWC);
      C = g.Cout; H = g.OH; W = g.OW;
    }
    h->cnn = true;
    h->cP = H * W;
    h->F = C * H * W;
    nblk = 2;
  }
  // value_hidden_sizes != policy_hidden_sizes (same depth): cfg->hidden sizes the critics, cfg->policy_hidden the policy nets
  // ... and lists of different LENGTH (policy_n_hidden > 0): every layer-indexed table below is sized by the deeper family
  // (h->L), the stage lists / heads / tiles take each net's own depth (chain_L)
  int pol_hidden[DSACT_MAX_HIDDEN_LAYERS] = {0};
  bool unequal = false;
  h->Lq = cfg->n_hidden;
  h->Lp = cfg->policy_n_hidden > 0 ? cfg->policy_n_hidden : cfg->n_hidden;
  if (h->Lp > DSACT_MAX_HIDDEN_LAYERS) return fail(h, DSACT_E_INVALID, "policy_n_hidden must be 0 (= n_hidden) or 1..%d", DSACT_MAX_HIDDEN_LAYERS);
  if (h->Lp != h->Lq && cfg->conv_type != DSACT_CONV_NONE) return fail(h, DSACT_E_INVALID, "policy_n_hidden: the CNN nets' MLP widths are fixed by conv_type");
  unequal = h->Lp != h->Lq;
  h->L = h->Lp > h->Lq ? h->Lp : h->Lq;
  for (int l = 0; l < h->Lp; ++l) {
    pol_hidden[l] = cfg->policy_hidden[l] > 0 ? cfg->policy_hidden[l] : (l < h->Lq ? cfg->hidden[l] : 0);
    if (pol_hidden[l] < 1 || pol_hidden[l] > kMaxWidth) return fail(h, DSACT_E_INVALID, "policy hidden width of layer %d must be 1..%d", l, kMaxWidth);
    unequal = unequal || l >= h->Lq || pol_hidden[l] != cfg->hidden[l];
  }
  if (unequal && (cfg->conv_type != DSACT_CONV_NONE || cfg->algo != 0))
    return fail(h, DSACT_E_INVALID, "value_hidden_sizes != policy_hidden_sizes is built for DSAC_V2 with MLP nets (tile-stage kernels)");
  // policy_std_type "mlp_separated": the policy nets are twin trunks (NetDesc::nblk == 2) beside single-trunk critics -- their
  // activation rows are twice as wide as their hidden sizes, so this is an unequal-widths configuration too (tile-stage kernels)
  const int nblk_pol = cfg->policy_twin ? 2 : nblk;
  h->unequal_widths = unequal || nblk_pol != nblk;
  for (int l = 0; l < h->L; ++l) {
    h->wq[l] = l < h->Lq ? nblk * cfg->hidden[l] : 0; h->wp[l] = l < h->Lp ? nblk_pol * pol_hidden[l] : 0;
    h->w[l] = h->wq[l] > h->wp[l] ? h->wq[l] : h->wp[l];
  }
  for (int l = 0; l < h->L; ++l)
    if (h->w[l] > kMaxWidth) return fail(h, DSACT_E_INVALID, "activation row width %d exceeds %d", h->w[l], kMaxWidth);
  h->ldx = (h->F + h->A + 3) & ~3;
  h->use_w1p = (size_t)4 * nblk * cfg->hidden[0] * h->ldx <= ((size_t)4 << 20);
  build_net(h->qd, h->F + h->A, cfg->hidden, h->Lq, 2, nblk, h->n_conv, h->cg);
  build_net(h->pd, h->F, pol_hidden, h->Lp, 2 * h->A, nblk_pol, h->n_conv, h->cg);
  h->n_q = h->qd.count; h->n_pi = h->pd.count;
  if (cfg->algo != DSACT_ALGO_DSAC_V2 && cfg->algo != DSACT_ALGO_DSAC_V1) return fail(h, DSACT_E_INVALID, "algo must be 0 (DSAC_V2) or 1 (DSAC_V1)");
  h->nq = cfg->algo == DSACT_ALGO_DSAC_V1 ? 1 : 2;
  h->n_online = h->nq * h->n_q + h->n_pi + 1;
  h->n_target = h->nq * h->n_q + h->n_pi;
  h->n_heads_wg = (h->B + 3) / 4;
  h->n_loss_wg = (h->B + 3) / 4;  // one wave per sample
  h->loss_rows = 4;
  h->auto_std_sums = h->B > 1024;   // large batches: the std column is summed once, not by every wave
  if (const char* v = getenv("DSACT_TIMELINE_STAGE")) h->env_timeline_stage = v;
  h->env_no_merged_gather = getenv("DSACT_NO_MERGED_GATHER") != nullptr;
  h->env_no_adam_pack = getenv("DSACT_NO_ADAM_PACK") != nullptr;
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && v > 0) h->n_cu = v; (void)hipGetLastError(); }
  if (const char* v = getenv("DSACT_CONV_DW_NKT")) h->env_conv_dw_nkt = atoi(v);
  for (int j = 0; j < kMaxConv; ++j) h->conv_dw_nkt_l[j] = h->env_conv_dw_nkt;   // (0: the per-layer default, set below once the geometry is known)
  if (const char* v = getenv("DSACT_CONV_DW_NKT_L")) {
    int j = 0;
    for (const char* p = v; *p && j < kMaxConv; ++j) {
      const int n = atoi(p);
      if (n >= 1 && n <= 3) h->conv_dw_nkt_l[j] = n;
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  // default: two k-tiles per workgroup (the dY tile of a step is staged once for both) unless the layer has three -- a 2 + 1
  // split leaves the odd workgroups half as long (measured at type_2, batch 256, profiles/r04_convdw_nkt.txt: layers
  // 0 / 2 / 3 / 4 / 5 35.5 / 24.0 / 16.6 / 19.6 / 12.7 -> 33.3 / 20.4 / 15.5 / 19.0 / 12.3 us, layer 1 36.6 -> 44.2)
  for (int j = 0; j < h->n_conv; ++j)
    if (h->conv_dw_nkt_l[j] < 1 || h->conv_dw_nkt_l[j] > 3) h->conv_dw_nkt_l[j] = tiles_of(h->cg[j].K + 4, TN) == 3 ? 1 : 2;
  for (int j = h->n_conv; j < kMaxConv; ++j) h->conv_dw_nkt_l[j] = 1;
  h->env_no_pipe = getenv("DSACT_NO_PIPE") != nullptr;
  h->env_no_pipe_defer = getenv("DSACT_NO_PIPE_DEFER") != nullptr;
  h->env_no_bqt = getenv("DSACT_NO_BQT_MERGE") != nullptr;
  h->env_no_bqp = getenv("DSACT_NO_BQP_MERGE") != nullptr;
  h->env_no_pipe_warm = getenv("DSACT_PIPE_WARM") == nullptr;
  h->env_no_pipe_tagged = getenv("DSACT_NO_PIPE_TAGGED") != nullptr;
  if (const char* v = getenv("DSACT_PK_PAD")) h->env_pk_pad = atoi(v) > 0 && atoi(v) <= 64 ? atoi(v) : 0;
  if (const char* v = getenv("DSACT_PIPE_MAP")) h->env_pipe_map = v;
  h->env_no_conv_dx_mfma = getenv("DSACT_NO_CONV_DX_MFMA") != nullptr;
  h->env_no_conv_narrow9 = getenv("DSACT_NO_CONV_NARROW9") != nullptr;
  h->env_no_conv_fwd64 = getenv("DSACT_NO_CONV_FWD64") != nullptr;
  h->env_no_conv_fwd32x64 = getenv("DSACT_NO_CONV_FWD32X64") != nullptr;
  h->env_conv_dw_sb3 = getenv("DSACT_CONV_DW_SB3") != nullptr;
  // default: the 16-channel layers whose K + 4 columns make three LDS k-tiles (type_2 layer 1: 37.0 -> 30.3 us)
  for (int j = 1; j < h->n_conv; ++j)
    if (h->cg[j].Cout == 16 && tiles_of(h->cg[j].K + 4, TN) == 3) h->env_conv_dw_reg |= 1 << j;
  if (const char* v = getenv("DSACT_CONV_DW_REG")) h->env_conv_dw_reg = atoi(v);
  if (const char* v = getenv("DSACT_CONV_DW_REG_WGS")) h->env_conv_dw_reg_wgs = atoi(v) > 0 ? atoi(v) : 512;
  if (const char* v = getenv("DSACT_CONV_FWD64_MIN")) h->env_conv_fwd64_min = atoi(v);
  if (const char* v = getenv("DSACT_DCOL64_MIN_M")) h->env_dcol64_min_m = atoi(v);
  h->env_no_dcol_ident = getenv("DSACT_NO_DCOL_IDENT") != nullptr;
  h->env_no_fat_stage = getenv("DSACT_NO_FAT_STAGE") != nullptr;
  if (const char* v = getenv("DSACT_CHAIN_RG")) h->env_chain_rg = atoi(v) == 1 ? 1 : atoi(v) == 4 ? 4 : 2;
  h->dw_chunks = (h->B > 448 && h->B % 256 == 0 && getenv("DSACT_NO_SPLITK") == nullptr) ? h->B / 256 : 1;   // (chain path: below)
  h->dw_part_stride = (h->n_online + 2 + 63) & ~(size_t)63;
  {
    // row-slice fused chains: MLP nets of DSAC_V2 with equal hidden widths of 64 / 128 / 256, batch a multiple of 16
    const int R = 4 * h->cRG;
    // (CNN nets: the twin MLP trunks over the conv features run as chain units -- DSACT_NO_CHAIN_CNN keeps them on the tile path)
    bool ok = h->B % R == 0 && h->B % 16 == 0 && (h->B <= 256 || h->B % 256 == 0) &&   // (round 6: any observation width -- the packed copies place a Q net's action columns element-wise when F % 4 != 0)
              getenv("DSACT_NO_CHAIN") == nullptr && !(h->nq == 1 && getenv("DSACT_NO_CHAIN_V1") != nullptr) &&
              !(h->cnn && (getenv("DSACT_NO_CHAIN_CNN") != nullptr || h->B > 1024));
    ok = ok && h->L <= kChMaxL;
    // (round 6: non-linear OUTPUT activations run on the chains too -- the generic-activation instantiations of the forward
    //  kernels carry them, the backward row phases multiply by their derivative; the throughput-regime kernels of batch >= 1024
    //  do not: `fat` below)
    ok = ok && !h->unequal_widths;                                     // one width per layer for every chain unit
    // a GELU OUTPUT layer: its derivative needs the pre-activation, which only the tile-stage heads keep (HeadsArgs::qdmean / pi_dact)
    ok = ok && cfg->value_out_act != OUT_ACT_GELU && cfg->policy_out_act != OUT_ACT_GELU;
    for (int l = 0; l < h->L; ++l) ok = ok && cfg->hidden[l] == cfg->hidden[0];
    const int W0 = cfg->hidden[0];
    ok = ok && (W0 == 64 || W0 == 128 || W0 == 256);
    h->s_obs = roundup((h->F + 3) / 4, kPD);
    h->s_act = roundup((h->A + 3) / 4, kPD);
    h->SoT = roundup((2 * h->A + 3) / 4, kPD);
    // LDS: input slice + two hidden slices + partial tiles must fit beside nothing else (one workgroup per CU)
    ok = ok && (size_t)chain_lds(4 * (h->s_obs + h->s_act), W0, R).total * sizeof(float) <= 150 * 1024;
    h->chain_ok = ok;
    h->twin = ok && h->cnn;
    if (h->cnn) h->env_pk_pad = 0;
    h->env_twin_seq = getenv("DSACT_TWIN_SEQ") != nullptr;
    if (ok) {
      // chain path: a weight-gradient tile contracts up to 1024 batch rows itself (rounds of 256, dw2_tile), so up to
      // batch 1024 there is ONE gradient arena and the optimiser stays fused into the tiles (no split-K partials, no
      // streaming Adam pass, and the graph keeps the riding gather); beyond that one partial arena per 1024 rows
      const int per = (h->B > 1024 && h->B % 1024 == 0 && getenv("DSACT_DW_RANGE_256") == nullptr) ? 1024 : 256;
      h->dw_chunks = h->B > 1024 || (h->B > 256 && getenv("DSACT_DW_RANGE_256") != nullptr) ? h->B / per : 1;
    }
    h->cW = W0; h->cNT = W0 / 64; h->n_slices = h->B / R;
    h->rg4_ok = ok && h->B % 16 == 0 && getenv("DSACT_NO_RG4") == nullptr &&
                (size_t)chain_lds(4 * (h->s_obs + h->s_act), W0, 16).total * sizeof(float) <= 150 * 1024;
    // merged forward launch: both groups resident at once (4-row group-B workgroups: batch <= 256), one flag per slice
    h->c_obs = (h->F + 15) / 16; h->c_act = (h->A + 15) / 16; h->c_out = (2 * h->A + 15) / 16;
    {
      const char* fm = getenv("DSACT_FAT_MIN");
      const int fat_min = fm ? atoi(fm) : 1024;
      const char* fb = getenv("DSACT_FAT_BWD_MIN");
      const int fat_bwd_min = fb ? atoi(fb) : 4096;
      // (the throughput-regime kernels hold DSAC_V2's two-critic row phase: one critic keeps the 8-row chains at every batch)
      h->fat = ok && !h->cnn && h->nq == 2 && h->B >= fat_min && h->B % 32 == 0 && (W0 == 128 || W0 == 256) && getenv("DSACT_NO_FAT") == nullptr &&
               cfg->value_out_act == 0 && cfg->policy_out_act == 0;   // (their heads are linear-only)
      h->fat_bwd = h->fat && h->B >= fat_bwd_min;
      if (const char* v = getenv("DSACT_FAT_RT")) h->env_fat_rt = atoi(v) == 2 ? 2 : 1;
    }
    h->fwd_merge = ok && !h->cnn && h->B <= 256 && h->B / 4 <= kChainFlagSlices && chain_rg(h, 4) == 1 && getenv("DSACT_NO_FWD_MERGE") == nullptr;
    h->pi_merge = ok && !h->cnn && h->B <= 512 && !h->fat_bwd && getenv("DSACT_NO_PI_MERGE") == nullptr;
  }
  Carver c0;
  carve(h, c0);
  h->ws_bytes = c0.off + 256;
  HIPCHK(h, hipMalloc(&h->ws, h->ws_bytes));
  HIPCHK(h, hipMemset(h->ws, 0, h->ws_bytes));
  Carver c1;
  c1.base = h->ws;
  carve(h, c1);
  {
    DevState st0;
    memset(&st0, 0, sizeof(st0));
    st0.b1p_q = st0.b2p_q = st0.b1p_pi = st0.b2p_pi = st0.b1p_alpha = st0.b2p_alpha = 1.0;
    HIPCHK(h, hipMemcpy(h->st, &st0, sizeof(st0), hipMemcpyHostToDevice));
  }
  {
    std::vector<float> ones(h->B, 1.0f);
    HIPCHK(h, hipMemcpy(h->ones, ones.data(), h->B * sizeof(float), hipMemcpyHostToDevice));
  }
  if (h->cnn) {
    for (int j = 0; j < h->n_conv; ++j) {
      const ConvGeom& g = h->cg[j];
      // patch origins are 32-bit float offsets, pixel indices go through an exact float-reciprocal division
      if ((size_t)h->Brows * g.H * g.W * g.Cin > 2147483647ull || (size_t)h->Brows * g.OH * g.OW >= (1u << 24))
        return fail(h, DSACT_E_INVALID, "batch x image too large for the conv index arithmetic (layer %d)", j);
    }
    std::vector<int> iota(h->Brows);
    for (int i = 0; i < h->Brows; ++i) iota[i] = i;
    HIPCHK(h, hipMemcpy(h->idx_iota, iota.data(), iota.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->own_stream = true;
  HIPCHK(h, hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  for (int j = 0; j <= kMaxConv; ++j) HIPCHK(h, hipEventCreateWithFlags(&h->ev_conv[j], hipEventDisableTiming));
  h->conv_fork = h->cnn && getenv("DSACT_CONV_FORK") != nullptr;
  if (h->conv_fork) for (int j = 0; j < kMaxConv; ++j) h->conv_dw_nkt_l[j] = 1;   // (the second-queue launches are the one-k-tile form)
  h->use_fork = getenv("DSACT_FORK") != nullptr && !h->cnn && h->dw_chunks == 1;  // measured: a forked graph branch costs +20 us/update (cross-queue signals) -> opt-in only
  {
    const int max_lds = (int)tile_lds_bytes(BK * kMaxPrefetchTiles);  // 129 KB of the CU's 160 KB
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, false, EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, false, EPI_GELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, false, EPI_GELU, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, true, EPI_MULG, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, true, EPI_MULG, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, true, EPI_MULG>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage_table, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_heads_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_heads_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_heads_bwd<3>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_heads_bwd<4>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage<false, true, EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_conv_dw<3>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<2, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd2<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd2<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwd2<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdpb<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdpb<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdpb<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdpb<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdpb<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdpb<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdt<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdt<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdt<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdt<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdt<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdt<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdp<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdp<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdp<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdp<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdp<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_fwdp<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_q<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_chain_bwd_pi<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage64<false, EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile64_lds_bytes()));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage64<true, EPI_MULG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile64_lds_bytes()));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_stage64<true, EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile64_lds_bytes()));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_conv_fwd64<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile64_lds_bytes()));
    HIPCHK(h, hipFuncSetAttribute((const void*)k_conv_fwd64<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile64_lds_bytes()));
  }
  HIPCHK(h, hipHostMalloc((void**)&h->handoff_host, 1024, hipHostMallocMapped));
  memset(h->handoff_host, 0, 1024);
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->handoff_dev, h->handoff_host, 0));
  h->act_out_host = (unsigned long long*)(h->handoff_host + 32); h->act_out_dev = (unsigned long long*)(h->handoff_dev + 32);
  HIPCHK(h, hipMalloc((void**)&h->act_h, (size_t)kActMaxLayers * kMaxWidth * sizeof(unsigned long long)));
  HIPCHK(h, hipMemset(h->act_h, 0, (size_t)kActMaxLayers * kMaxWidth * sizeof(unsigned long long)));
  h->env_no_fast_act = getenv("DSACT_NO_FAST_ACT") != nullptr;
  h->host_act = getenv("DSACT_NO_HOST_ACT") == nullptr;

  if (h->fwd_merge) {
    // the merged forward is sized for both groups' workgroups being resident at once: two per CU (speed, not
    // correctness -- consumers only wait for lower block ids, which are always dispatched first)
    int per_cu = 0;
    const size_t la = (size_t)chain_lds(4 * (h->s_obs + h->s_act), h->cW, 4 * chain_rg(h, 6)).total * sizeof(float);
    const void* fn = h->cNT == 1 ? (const void*)k_chain_fwd2<1> : h->cNT == 2 ? (const void*)k_chain_fwd2<2> : (const void*)k_chain_fwd2<4>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * h->cNT, la) != hipSuccess || per_cu < 2) h->fwd_merge = false;
    (void)hipGetLastError();
  }
  for (int i = 0; i < 8; ++i) {
    HIPCHK(h, hipHostMalloc((void**)&h->h_idx[i], (size_t)h->B * sizeof(int), hipHostMallocDefault));
    HIPCHK(h, hipEventCreateWithFlags(&h->h_idx_ev[i], hipEventDisableTiming));
  }
  return DSACT_OK;
}

int dsact_destroy(dsact_handle* h) {
  if (!h) return DSACT_E_INVALID;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  drop_graphs(h);
  if (h->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  for (auto& r : h->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  if (h->tev0) { hipEventDestroy(h->tev0); hipEventDestroy(h->tev1); }
  if (h->uev0) { hipEventDestroy(h->uev0); hipEventDestroy(h->uev1); }
  for (int i = 0; i < 8; ++i) {
    if (h->h_idx[i]) hipHostFree(h->h_idx[i]);
    if (h->h_idx_ev[i]) hipEventDestroy(h->h_idx_ev[i]);
  }
  if (h->handoff_host) hipHostFree(h->handoff_host);
  if (h->act_h) hipFree(h->act_h);
  if (h->pol_host) hipHostFree(h->pol_host);
  if (h->pol_ev) hipEventDestroy(h->pol_ev);
  free(h->act_buf);
  delete h->act_pool;
  if (h->d_tiles) hipFree(h->d_tiles);
  if (h->alt.d_tiles) hipFree(h->alt.d_tiles);
  if (h->alt_ws) hipFree(h->alt_ws);
  if (h->pipe_ws) hipFree(h->pipe_ws);
  if (h->bqt_cnt) hipFree(h->bqt_cnt);
  if (h->bqt_tab) hipFree(h->bqt_tab);
  for (int i = 0; i < 2; ++i) if (h->bqp_pairs[i]) hipFree(h->bqp_pairs[i]);
  if (h->pk_ws) hipFree(h->pk_ws);
  for (int i = 0; i < 2; ++i) { if (h->d_fwdt[i]) hipFree(h->d_fwdt[i]); delete h->fwdt_host[i]; }
  if (h->d_mir) hipFree(h->d_mir);
  if (h->d_apjobs) hipFree(h->d_apjobs);
  if (h->d_pack) hipFree(h->d_pack);
  if (h->idx_table) hipFree(h->idx_table);
  if (h->noise_table) hipFree(h->noise_table);
  for (int i = 0; i < 4; ++i) {
    if (h->grp_pin[i]) hipHostFree(h->grp_pin[i]);
    if (h->grp_ev[i]) hipEventDestroy(h->grp_ev[i]);
  }
  if (h->stage_dev) hipFree(h->stage_dev);
  for (int i = 0; i < 2; ++i) {
    if (h->stage_pin[i]) hipHostFree(h->stage_pin[i]);
    if (h->stage_ev[i]) hipEventDestroy(h->stage_ev[i]);
  }
  if (h->logp_stage) hipFree(h->logp_stage);
  if (h->stage_img) hipFree(h->stage_img);
  for (float* p : {h->rb_obs, h->rb_obs2, h->rb_act, h->rb_rew, h->rb_done, h->rb_logp})
    if (p) hipFree(p);
  if (h->ws) hipFree(h->ws);
  if (h->aux_stream) { hipStreamSynchronize(h->aux_stream); hipStreamDestroy(h->aux_stream); }
  for (int j = 0; j <= kMaxConv; ++j) if (h->ev_conv[j]) hipEventDestroy(h->ev_conv[j]);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
  return DSACT_OK;
}

int dsact_set_stream(dsact_handle* h, void* s) {
  if (!h) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (any_graph(h)) return fail(h, DSACT_E_STATE, "cannot change stream after dsact_graph_build");
  if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->own_stream && h->stream) { hipStreamDestroy(h->stream); h->stream = nullptr; }
  if (s) { h->stream = (hipStream_t)s; h->own_stream = false; }
  else { HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
  return DSACT_OK;
}

void* dsact_get_stream(const dsact_handle* h) { return h ? (void*)h->stream : nullptr; }

int dsact_sync(dsact_handle* h) {
  if (!h) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_handoff(h);
}

size_t dsact_online_count(const dsact_handle* h) { return h ? h->n_online : 0; }
size_t dsact_target_count(const dsact_handle* h) { return h ? h->n_target : 0; }
size_t dsact_q_count(const dsact_handle* h) { return h ? h->n_q : 0; }
size_t dsact_pi_count(const dsact_handle* h) { return h ? h->n_pi : 0; }

int dsact_bind_arenas(dsact_handle* h, float* online, float* target, float* adam_m, float* adam_v, float* grads) {
  if (!h) return DSACT_E_INVALID;
  if (!online || !target || !adam_m || !adam_v || !grads) return fail(h, DSACT_E_INVALID, "null arena pointer");
  HIPCHK(h, hipSetDevice(h->device));
  if (any_graph(h)) return fail(h, DSACT_E_STATE, "cannot rebind arenas after dsact_graph_build");
  h->online = online; h->target = target; h->adam_m = adam_m; h->adam_v = adam_v; h->grads = grads;
  TRY(build_chain(h));
  TRY(build_pack_jobs(h));
  if (h->chain_ok && !h->twin) TRY(build_adam_pack_jobs(h));
  TRY(build_tasks(h));
  if (h->twin) TRY(build_twin_fwd(h));
  if (h->chain_ok) (void)dw2_args(h, false);   // tile ranges of the dw2 problem list
  if (!h->cnn) {   // the same task lists over the second batch set
    TRY(alloc_alt_set(h));
    select_set(h, 1);
    int rc = build_tasks(h);
    select_set(h, 0);
    TRY(rc);
  }
  h->state_invalid = false;
  h->pol_epoch++;   // (new arenas: the host-side acting snapshot is stale; the next acting call copies again)
  return DSACT_OK;
}

int dsact_set_action_limits(dsact_handle* h, const float* high, const float* low) {
  if (!h || !high || !low) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<float> s(h->A), c(h->A);
  for (int j = 0; j < h->A; ++j) {
    if (!(high[j] > low[j])) return fail(h, DSACT_E_INVALID, "action_high_limit must exceed action_low_limit");
    s[j] = (high[j] - low[j]) / 2;  // fp32 like the reference's tensor arithmetic
    c[j] = (high[j] + low[j]) / 2;
    if (h->cfg.act_dist == 1) { s[j] = 0.0f; c[j] = 0.0f; }   // GaussDistribution: the closed forms' "no squashing" selector (dsact_math.h)
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h->act_scale, s.data(), h->A * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->act_center, c.data(), h->A * sizeof(float), hipMemcpyHostToDevice));
  for (int j = 0; j < h->A && j < 32; ++j) { h->act_scale_h[j] = s[j]; h->act_center_h[j] = c[j]; }
  h->limits_set = true;
  return DSACT_OK;
}

int dsact_get_state(dsact_handle* h, int32_t adam_steps[3], float mean_std[2]) {
  if (!h) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  DevState st;
  HIPCHK(h, hipMemcpy(&st, h->st, sizeof(st), hipMemcpyDeviceToHost));
  TRY(check_handoff(h));
  if (adam_steps) { adam_steps[0] = st.t_q; adam_steps[1] = st.t_pi; adam_steps[2] = st.t_alpha; }
  if (mean_std) { mean_std[0] = st.ms_init ? st.ms1 : -1.0f; mean_std[1] = st.ms_init ? st.ms2 : -1.0f; }
  return DSACT_OK;
}

int dsact_set_state(dsact_handle* h, const int32_t adam_steps[3], const float mean_std[2]) {
  if (!h) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  DevState st;
  HIPCHK(h, hipMemcpy(&st, h->st, sizeof(st), hipMemcpyDeviceToHost));
  if (adam_steps) {
    st.t_q = adam_steps[0]; st.t_pi = adam_steps[1]; st.t_alpha = adam_steps[2];
    const double b1 = h->cfg.adam_beta1, b2 = h->cfg.adam_beta2;
    st.b1p_q = pow(b1, st.t_q); st.b2p_q = pow(b2, st.t_q);
    st.b1p_pi = pow(b1, st.t_pi); st.b2p_pi = pow(b2, st.t_pi);
    st.b1p_alpha = pow(b1, st.t_alpha); st.b2p_alpha = pow(b2, st.t_alpha);
  }
  if (mean_std) {
    st.ms_init = (mean_std[0] >= 0.0f && mean_std[1] >= 0.0f) ? 1 : 0;
    st.ms1 = mean_std[0]; st.ms2 = mean_std[1];
  }
  HIPCHK(h, hipMemcpy(h->st, &st, sizeof(st), hipMemcpyHostToDevice));
  // a FULL restore (step counters and the EMA, after the arenas were rewritten) acknowledges a hand-over timeout; a partial
  // call (e.g. only mean_std tweaked) does not: dsact_debug_set("ack_state", 1) is the explicit acknowledgement
  if (adam_steps && mean_std) h->state_invalid = false;
  return DSACT_OK;
}

// ---- replay ring ---------------------------------------------------------------------------------
int dsact_set_hyper(dsact_handle* h, int32_t which, double value) {
  if (!h) return DSACT_E_INVALID;
  if (!(value == value)) return fail(h, DSACT_E_INVALID, "hyper-parameter is NaN");
  switch (which) {
    case DSACT_HYPER_GAMMA: h->cfg.gamma = value; break;
    case DSACT_HYPER_TAU:
      if (value < 0.0 || value > 1.0) return fail(h, DSACT_E_INVALID, "tau outside [0, 1]");
      h->cfg.tau = value; break;
    case DSACT_HYPER_TAU_B:
      if (value < 0.0 || value > 1.0) return fail(h, DSACT_E_INVALID, "tau_b outside [0, 1]");
      h->cfg.tau_b = value; break;
    case DSACT_HYPER_AUTO_ALPHA: h->cfg.auto_alpha = value != 0.0 ? 1 : 0; break;
    case DSACT_HYPER_ALPHA: h->cfg.alpha_fixed = value; break;
    case DSACT_HYPER_DELAY_UPDATE:
      if (value < 1.0 || value != (double)(int32_t)value) return fail(h, DSACT_E_INVALID, "delay_update must be a positive integer");
      h->cfg.delay_update = (int32_t)value; break;
    case DSACT_HYPER_TD_BOUND:
      if (h->cfg.algo != 1) return fail(h, DSACT_E_INVALID, "TD_bound is a DSAC_V1 parameter");
      h->cfg.td_bound = value; break;
    case DSACT_HYPER_V1_BOUND:
      if (h->cfg.algo != 1) return fail(h, DSACT_E_INVALID, "bound is a DSAC_V1 parameter");
      h->cfg.v1_unbounded = value != 0.0 ? 0 : 1; break;
    default: return fail(h, DSACT_E_INVALID, "unknown hyper-parameter %d", (int)which);
  }
  // every launch reads h->cfg when it is enqueued; only a captured graph holds old values
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  drop_graphs(h);
  return DSACT_OK;
}

int dsact_buffer_create(dsact_handle* h, int64_t capacity) {
  if (!h || capacity < 1) return DSACT_E_INVALID;
  if (capacity > 2147483647LL) return fail(h, DSACT_E_INVALID, "capacity must fit int32 (device indices)");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->rb_obs) return fail(h, DSACT_E_STATE, "buffer already created");
  const size_t N = (size_t)capacity;
  HIPCHK(h, hipMalloc(&h->rb_obs, N * h->O * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->rb_obs2, N * h->O * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->rb_act, N * h->A * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->rb_rew, N * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->rb_done, N * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->rb_logp, N * sizeof(float)));
  // replay_buffer.py:25-38: zero-initialised
  HIPCHK(h, hipMemsetAsync(h->rb_obs, 0, N * h->O * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->rb_obs2, 0, N * h->O * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->rb_act, 0, N * h->A * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->rb_rew, 0, N * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->rb_done, 0, N * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->rb_logp, 0, N * sizeof(float), h->stream));
  h->cap = capacity; h->ptr = 0; h->size = 0;
  return DSACT_OK;
}

int64_t dsact_buffer_size(const dsact_handle* h) { return h ? h->size : -1; }
int64_t dsact_buffer_ptr(const dsact_handle* h) { return h ? h->ptr : -1; }

int dsact_buffer_add(dsact_handle* h, int64_t n, const float* obs, const float* act, const float* rew,
                     const float* obs2, const float* done, const float* logp) {
  if (!h || n < 0 || !obs || !act || !rew || !obs2 || !done) return DSACT_E_INVALID;
  if (!h->rb_obs) return fail(h, DSACT_E_STATE, "buffer not created");
  if (n == 0) return DSACT_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t O = h->O, A = h->A;
  const size_t row_f = 2 * O + A + 3;
  if ((size_t)n > h->stage_rows) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->stage_dev) hipFree(h->stage_dev);
    for (int i = 0; i < 2; ++i)
      if (h->stage_pin[i]) { hipHostFree(h->stage_pin[i]); h->stage_pin[i] = nullptr; }
    h->stage_rows = (size_t)n < 64 ? 64 : (size_t)n;
    HIPCHK(h, hipMalloc(&h->stage_dev, h->stage_rows * row_f * sizeof(float)));
    for (int i = 0; i < 2; ++i) {
      HIPCHK(h, hipHostMalloc((void**)&h->stage_pin[i], h->stage_rows * row_f * sizeof(float), hipHostMallocDefault));
      if (!h->stage_ev[i]) HIPCHK(h, hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
    }
  }
  // The caller's arrays are copied into a pinned slot here (they may be reused right away); the H2D copy and the ring
  // write are stream-ordered behind whatever the handle is doing and nothing waits for them: the sampler's next
  // env.step overlaps the transfer. A slot is reused two calls later, after its copy has left it.
  const int slot = (int)(h->stage_k++ & 1);
  HIPCHK(h, hipEventSynchronize(h->stage_ev[slot]));
  float* pin = h->stage_pin[slot];
  size_t off = 0;
  auto put = [&](const float* src, size_t count) {
    if (src) memcpy(pin + off, src, count * sizeof(float));
    off += count;
  };
  put(obs, (size_t)n * O); put(obs2, (size_t)n * O); put(act, (size_t)n * A); put(rew, (size_t)n); put(done, (size_t)n);
  put(logp, (size_t)n);   // absent: the ring's logp column keeps its old values (a.s_logp == nullptr below)
  const size_t R = (size_t)n;   // the staged columns are packed for THIS call's n
  float* s_obs = h->stage_dev;
  float* s_obs2 = s_obs + R * O;
  float* s_act = s_obs2 + R * O;
  float* s_rew = s_act + R * A;
  float* s_done = s_rew + R;
  float* s_logp = s_done + R;
  HIPCHK(h, hipMemcpyAsync(h->stage_dev, pin, off * sizeof(float), hipMemcpyHostToDevice, h->stream));   // ONE copy
  HIPCHK(h, hipEventRecord(h->stage_ev[slot], h->stream));
  // if n exceeds the capacity only the last `cap` rows survive (same as sequential store())
  long long first = 0, cnt = n;
  if (n > h->cap) { first = n - h->cap; cnt = h->cap; }
  ScatterArgs a;
  a.s_obs = s_obs + first * O; a.s_obs2 = s_obs2 + first * O; a.s_act = s_act + first * A;
  a.s_rew = s_rew + first; a.s_done = s_done + first; a.s_logp = logp ? s_logp + first : nullptr;
  a.rb_obs = h->rb_obs; a.rb_obs2 = h->rb_obs2; a.rb_act = h->rb_act; a.rb_rew = h->rb_rew; a.rb_done = h->rb_done; a.rb_logp = h->rb_logp;
  a.ptr = (h->ptr + first) % h->cap; a.cap = h->cap; a.n = (int)cnt; a.O = h->O; a.A = h->A;
  if (h->cnn) {
    // image rows are wide (C*H*W floats): a grid of blocks per row for the two images, the wave-per-row
    // kernel for the narrow columns
    ImgScatterArgs w;
    w.s_obs = a.s_obs; w.s_obs2 = a.s_obs2; w.rb_obs = h->rb_obs; w.rb_obs2 = h->rb_obs2;
    w.ptr = a.ptr; w.cap = h->cap; w.n = (int)cnt; w.O = h->O;
    TRY(launch(h, "ring_write_img", k_ring_write_img, dim3(8, (unsigned)cnt), dim3(kThreads), 0, w));
    a.O = 0;
  }
  TRY(launch(h, "ring_write", k_ring_write, dim3((unsigned)((cnt + 3) / 4)), dim3(kThreads), 0, a));
  h->ptr = (h->ptr + n) % h->cap;
  h->size = h->size + n > h->cap ? h->cap : h->size + n;
  return DSACT_OK;
}

int dsact_buffer_fill_device(dsact_handle* h, int64_t row0, int64_t n, const float* obs, const float* act,
                             const float* rew, const float* obs2, const float* done) {
  if (!h || row0 < 0 || n < 0) return DSACT_E_INVALID;
  if (!h->rb_obs) return fail(h, DSACT_E_STATE, "buffer not created");
  if (row0 + n > h->cap) return fail(h, DSACT_E_INVALID, "fill exceeds capacity");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t O = h->O, A = h->A;
  HIPCHK(h, hipMemcpyAsync(h->rb_obs + row0 * O, obs, n * O * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->rb_obs2 + row0 * O, obs2, n * O * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->rb_act + row0 * A, act, n * A * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->rb_rew + row0, rew, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->rb_done + row0, done, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (row0 + n > h->size) h->size = row0 + n;
  h->ptr = (row0 + n) % h->cap;
  return DSACT_OK;
}

int dsact_gather(dsact_handle* h, const int64_t* idx_host, int32_t batch) {
  if (!h || !idx_host) return DSACT_E_INVALID;
  if (!h->rb_obs) return fail(h, DSACT_E_STATE, "buffer not created");
  if (batch != h->B) return fail(h, DSACT_E_INVALID, "batch %d != configured batch %d", batch, h->B);
  if (h->size == 0) return fail(h, DSACT_E_STATE, "buffer empty");
  HIPCHK(h, hipSetDevice(h->device));
  const int slot = h->h_idx_slot;
  h->h_idx_slot = (slot + 1) & 7;
  HIPCHK(h, hipEventSynchronize(h->h_idx_ev[slot]));
  for (int i = 0; i < batch; ++i) {
    if (idx_host[i] < 0 || idx_host[i] >= h->size) return fail(h, DSACT_E_INVALID, "index %lld out of range [0,%lld)", (long long)idx_host[i], h->size);
    h->h_idx[slot][i] = (int)idx_host[i];
  }
  HIPCHK(h, hipMemcpyAsync(h->idx_eager, h->h_idx[slot], batch * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->h_idx_ev[slot], h->stream));
  // bookkeeping (iteration, counters) is done by the step call; this gather only stages rows
  if (h->cnn) {
    TRY(enqueue_gather_img(h, h->rb_obs, h->rb_obs2, h->idx_eager, 1, 0, true, h->img[0], h->img[1], h->B));
    h->have_batch = true;
    return DSACT_OK;
  }
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.rb_obs = h->rb_obs; a.rb_obs2 = h->rb_obs2; a.rb_act = h->rb_act; a.rb_rew = h->rb_rew; a.rb_done = h->rb_done;
  a.idx_table = h->idx_eager; a.idx_rows = 1; a.use_dev = 0; a.host_it = 0; a.host_row = 0;
  a.X0 = h->X0; a.XP = h->XP; a.X2 = h->X2; a.rew = h->rew; a.done = h->done;
  a.B = h->B; a.O = h->O; a.A = h->A; a.ldx = h->ldx; a.st = h->st; a.bookkeeping = 0; a.advance_counters = 0;
  a.hp = step_hyper(h); a.nz = noise_args(h); a.nz.seed = 0;
  a.n_gather_blocks = (h->B + 3) / 4;
  a.rp = repack_args(h, 0);
  TRY(launch(h, "gather", k_gather, dim3((h->B + 3) / 4), dim3(kThreads), 0, a));
  h->have_batch = true;
  return DSACT_OK;
}

int dsact_read_batch(dsact_handle* h, float* obs, float* act, float* rew, float* obs2, float* done, float* logp) {
  if (!h) return DSACT_E_INVALID;
  if (!h->have_batch) return fail(h, DSACT_E_STATE, "no minibatch staged");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t B = h->B, O = h->O, A = h->A, ld = h->ldx;
  if (h->cnn) {
    // staged images are pixel-major; the caller gets (C,H,W) rows back (host-side permutation: slow path)
    const size_t C = h->cfg.img_c, HW = (size_t)h->cfg.img_h * h->cfg.img_w;
    std::vector<float> tmp(B * O);
    for (int w = 0; w < 2; ++w) {
      float* dst = w == 0 ? obs : obs2;
      if (!dst) continue;
      HIPCHK(h, hipMemcpy(tmp.data(), h->img[w], B * O * 4, hipMemcpyDeviceToHost));
      for (size_t b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p)
          for (size_t c = 0; c < C; ++c) dst[b * O + c * HW + p] = tmp[b * O + p * C + c];
    }
    if (act) HIPCHK(h, hipMemcpy2D(act, A * 4, h->Xc[C_Q1C] + h->F, ld * 4, A * 4, B, hipMemcpyDeviceToHost));
  } else {
    if (obs) HIPCHK(h, hipMemcpy2D(obs, O * 4, h->X0, ld * 4, O * 4, B, hipMemcpyDeviceToHost));
    if (obs2) HIPCHK(h, hipMemcpy2D(obs2, O * 4, h->X2, ld * 4, O * 4, B, hipMemcpyDeviceToHost));
    if (act) HIPCHK(h, hipMemcpy2D(act, A * 4, h->X0 + O, ld * 4, A * 4, B, hipMemcpyDeviceToHost));
  }
  if (rew) HIPCHK(h, hipMemcpy(rew, h->rew, B * 4, hipMemcpyDeviceToHost));
  if (done) HIPCHK(h, hipMemcpy(done, h->done, B * 4, hipMemcpyDeviceToHost));
  if (logp) {
    if (!h->rb_logp) return fail(h, DSACT_E_STATE, "buffer not created");
    // one gather launch + ONE device-to-host copy (was one 4-byte copy per row)
    if (!h->logp_stage) HIPCHK(h, hipMalloc(&h->logp_stage, B * sizeof(float)));
    TakeArgs t;
    t.src = h->rb_logp; t.idx = h->idx_eager; t.dst = h->logp_stage; t.n = (int)B;
    TRY(launch(h, "take_logp", k_take, dim3((unsigned)((B + kThreads - 1) / kThreads)), dim3(kThreads), 0, t));
    HIPCHK(h, hipMemcpyAsync(logp, h->logp_stage, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return DSACT_OK;
}

int dsact_load_batch(dsact_handle* h, const float* obs, const float* act, const float* rew, const float* obs2, const float* done) {
  if (!h || !obs || !act || !rew || !obs2 || !done) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t B = h->B, O = h->O, A = h->A, ld = h->ldx;
  hipStream_t s = h->stream;
  if (h->cnn) {
    const size_t R = h->Brows;
    if (!h->stage_img) HIPCHK(h, hipMalloc(&h->stage_img, 2 * R * O * sizeof(float)));
    HIPCHK(h, hipMemcpyAsync(h->stage_img, obs, B * O * 4, hipMemcpyDefault, s));
    HIPCHK(h, hipMemcpyAsync(h->stage_img + R * O, obs2, B * O * 4, hipMemcpyDefault, s));
    TRY(enqueue_gather_img(h, h->stage_img, h->stage_img + R * O, h->idx_iota, 1, 0, false, h->img[0], h->img[1], h->B));
    HIPCHK(h, hipMemcpy2DAsync(h->Xc[C_Q1C] + h->F, ld * 4, act, A * 4, A * 4, B, hipMemcpyDefault, s));
    HIPCHK(h, hipMemcpy2DAsync(h->Xc[C_Q2C] + h->F, ld * 4, act, A * 4, A * 4, B, hipMemcpyDefault, s));
    HIPCHK(h, hipMemcpyAsync(h->rew, rew, B * 4, hipMemcpyDefault, s));
    HIPCHK(h, hipMemcpyAsync(h->done, done, B * 4, hipMemcpyDefault, s));
    HIPCHK(h, hipStreamSynchronize(s));
    h->have_batch = true;
    return DSACT_OK;
  }
  HIPCHK(h, hipMemcpy2DAsync(h->X0, ld * 4, obs, O * 4, O * 4, B, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpy2DAsync(h->XP, ld * 4, h->X0, ld * 4, O * 4, B, hipMemcpyDeviceToDevice, s));  // obs crosses the bus once
  HIPCHK(h, hipMemcpy2DAsync(h->X2, ld * 4, obs2, O * 4, O * 4, B, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpy2DAsync(h->X0 + O, ld * 4, act, A * 4, A * 4, B, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(h->rew, rew, B * 4, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(h->done, done, B * 4, hipMemcpyDefault, s));
  HIPCHK(h, hipStreamSynchronize(s));  // the source buffers are the caller's
  h->have_batch = true;
  return DSACT_OK;
}

int dsact_upload_index_table(dsact_handle* h, const int64_t* idx_host, int32_t rows) {
  if (!h || !idx_host || rows < 1) return DSACT_E_INVALID;
  if (!h->rb_obs) return fail(h, DSACT_E_STATE, "buffer not created");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t n = (size_t)rows * h->B;
  std::vector<int> tmp(n);
  for (size_t i = 0; i < n; ++i) {
    if (idx_host[i] < 0 || idx_host[i] >= h->size) return fail(h, DSACT_E_INVALID, "index out of range");
    tmp[i] = (int)idx_host[i];
  }
  if (any_graph(h) && rows != h->idx_rows) return fail(h, DSACT_E_STATE, "index table shape is baked into the captured graph");
  if (!h->idx_table || rows != h->idx_rows) {
    drop_graphs(h);   // (cached ones included: the table's address and row count are baked into them)
    if (h->idx_table) hipFree(h->idx_table);
    if (h->noise_table) { hipFree(h->noise_table); h->noise_table = nullptr; }   // sized by the row count
    HIPCHK(h, hipMalloc(&h->idx_table, n * sizeof(int)));
    h->idx_rows = rows;
  }
  HIPCHK(h, hipMemcpy(h->idx_table, tmp.data(), n * sizeof(int), hipMemcpyHostToDevice));
  return DSACT_OK;
}

// ---- noise ---------------------------------------------------------------------------------------
int dsact_set_noise(dsact_handle* h, const float* eps_new, const float* eps_2, const float* z5, const float* z6) {
  if (!h || !eps_new || !eps_2 || !z5 || !z6) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t B = h->B, A = h->A;
  HIPCHK(h, hipMemcpyAsync(h->eps_new, eps_new, B * A * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->eps_2, eps_2, B * A * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->z5, z5, B * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->z6, z6, B * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->rng_seed = 0;
  return DSACT_OK;
}

int dsact_set_device_rng(dsact_handle* h, uint64_t seed) {
  if (!h) return DSACT_E_INVALID;
  if (any_graph(h)) return fail(h, DSACT_E_STATE, "rng mode is baked into the captured graph");
  h->rng_seed = seed;
  return DSACT_OK;
}

// ---- update --------------------------------------------------------------------------------------
// device time of the last eager update (the reference's "Time/Algorithm time" is the wall time of a synchronous CPU
// update; here the call returns after the enqueue, so the honest figure is the stream time between these two events)
// ---- host-side acting (dsact_host_act.h) ------------------------------------------------------------------------------
// the policy net's parameters -> the pinned snapshot, stream-ordered behind everything enqueued so far
static int enqueue_policy_copy(dsact_handle* h) {
  HIPCHK(h, hipMemcpyAsync(h->pol_host, net_params(h, N_POL), (size_t)h->n_pi * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(h->pol_ev, h->stream));
  h->pol_pending = true;
  h->pol_copied = h->pol_epoch;
  h->act_copies++;
  return DSACT_OK;
}
// called by every entry point that enqueued work which may change the policy's parameters (an update with
// iteration % delay_update == 0, graph replays, the data-parallel apply): the snapshot of a handle that acts on the host is
// refreshed right behind it, so the copy is already in flight when the sampler asks
static int policy_moved(dsact_handle* h) {
  h->pol_epoch++;
  if (h->pol_host && h->host_act) return enqueue_policy_copy(h);
  return DSACT_OK;
}
static bool act_fast_ok(const dsact_handle* h);
static bool act_host_ok(const dsact_handle* h) { return h->host_act && act_fast_ok(h); }
// policy(obs) [+ TanhGaussDistribution.sample()] on the calling thread with the weights of the last completed update
// (training/off_sampler.py:46-56). eps == nullptr: out = the 2A logits; else out = A actions and *logp
static int act_forward_host(dsact_handle* h, const float* obs_host, const float* eps, float* out, float* logp) {
  TRY(check_handoff(h));
  const auto t0 = std::chrono::steady_clock::now();
  if (!h->pol_host) {
    HIPCHK(h, hipHostMalloc((void**)&h->pol_host, ((size_t)h->n_pi + 64) * sizeof(float), hipHostMallocDefault));
    HIPCHK(h, hipEventCreateWithFlags(&h->pol_ev, hipEventDisableTiming));
    h->act_buf = (float*)malloc((size_t)(2 * (kMaxWidth + 64) + 128) * sizeof(float));
    if (!h->act_buf) return fail(h, DSACT_E_HIP, "out of host memory");
  }
  if (h->pol_copied != h->pol_epoch) TRY(enqueue_policy_copy(h));
  if (h->pol_pending) {
    // poll first: a blocking wait sleeps the thread and its wake-up (tens of us) would sit in front of every burst of
    // environment steps; after ~5 ms fall back to the blocking wait
    hipError_t q = hipErrorNotReady;
    while ((q = hipEventQuery(h->pol_ev)) == hipErrorNotReady) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
    }
    if (q != hipSuccess) HIPCHK(h, hipEventSynchronize(h->pol_ev));
    h->pol_pending = false;
    TRY(check_handoff(h));     // (an update kernel that gave up while the stream drained: this snapshot is not to be acted on)
    h->act_copy_wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  hostact::Layer ly[kActMaxLayers];
  for (int l = 0; l <= h->Lp; ++l) {
    ly[l].W = h->pol_host + h->pd.w_off[l]; ly[l].b = h->pol_host + h->pd.b_off[l];
    ly[l].K = h->pd.in[l]; ly[l].N = h->pd.out[l]; ly[l].half = 0;
    if (h->pd.nblk == 2 && l > 0 && l < h->Lp) { ly[l].K = h->pd.in[l] / 2; ly[l].half = h->pd.out[l] / 2; }   // two (H x Hprev) blocks
  }
  float* b0 = h->act_buf; float* b1 = b0 + kMaxWidth + 64; float* raw = b1 + kMaxWidth + 64;
  if (h->act_threads == 0) {
    // How many threads share a wide layer (dsact_host_act.h, Pool; the result is bit-identical for every count). The weights
    // (0.95 MB at Humanoid 3x256) do not stay in one core's L2 between two environment steps, so one thread streams them from
    // the L3 (8.9 us per forward on the boxes' EPYC 9575F); four cores of ONE core complex hold a quarter each in their own L2.
    // Unpinned helpers measured no gain there (9.6 us: every fork-join crossed CCDs / sockets) and 2 threads were 4x SLOWER
    // than 1 in the 8-vCPU build container -- so helpers are used only where they can be pinned beside the calling thread's
    // core (same last-level cache, Linux sysfs), on hosts with >= 16 CPUs, for policies at least 128 wide.
    // DSACT_HOST_ACT_THREADS=n forces a count (1: never any helper).
    const char* ev = getenv("DSACT_HOST_ACT_THREADS");
    int t = 1;
    std::vector<int> pin;
#if defined(__linux__)
    const int cpu = sched_getcpu();
    if (cpu >= 0) pin = hostact::llc_sibling_cores(cpu);
#endif
    if (ev) t = atoi(ev);
    else if ((int)std::thread::hardware_concurrency() >= 16 && h->pd.out[0] >= 128 && pin.size() >= 3) t = 4;
    t = t < 1 ? 1 : (t > 16 ? 16 : t);
    if (t > 1) {
      h->act_pool = new hostact::Pool(t, pin);
      if (!ev && h->act_pool->pinned() < t - 1) { delete h->act_pool; h->act_pool = nullptr; t = 1; }   // (a cpuset refused the pinning)
    }
    h->act_threads = t;
#if defined(__linux__)
    h->act_pin = pin; h->act_cpu = cpu;
#endif
  }
#if defined(__linux__)
  // The scheduler may move the calling thread to another core complex (a long-running trainer, other legs of a benchmark in
  // the same process): helpers pinned beside the OLD core then make every layer's fork-join cross complexes (measured: 2.7 ->
  // 5.7 us per forward). Every 256th call: if the caller's CPU is neither where it was nor among the helpers' cache siblings,
  // the helpers follow it.
  if (h->act_pool && h->act_pool->pinned() > 0 && (h->act_host_calls & 255) == 255) {
    const int cpu = sched_getcpu();
    if (cpu >= 0 && cpu != h->act_cpu) {
      h->act_cpu = cpu;
      const std::vector<int> pin = hostact::llc_sibling_cores(cpu);   // (sysfs reads: only when the caller actually moved)
      const size_t need = (size_t)h->act_threads - 1;
      bool same = pin.size() >= need && h->act_pin.size() >= need;
      for (size_t i = 0; same && i < need; ++i) same = pin[i] == h->act_pin[i];
      if (!same && pin.size() >= need) { h->act_pool->repin(pin); h->act_pin = pin; h->act_repins++; }
    }
  }
#endif
  const auto t1 = std::chrono::steady_clock::now();
  hostact::forward(ly, h->Lp + 1, h->cfg.policy_act, obs_host, b0, b1, raw, h->act_pool);
  hostact::head(raw, h->A, h->cfg.min_log_std, h->cfg.max_log_std, eps, h->act_scale_h, h->act_center_h, out, logp,
                h->cfg.policy_out_act, h->cfg.policy_std_param ? h->A : 2 * h->A);
  h->act_host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
  h->act_host_calls++;
  return DSACT_OK;
}

static int mark_update(dsact_handle* h, bool end) {
  if (!h->uev0) {
    HIPCHK(h, hipEventCreate(&h->uev0));
    HIPCHK(h, hipEventCreate(&h->uev1));
  }
  HIPCHK(h, hipEventRecord(end ? h->uev1 : h->uev0, h->stream));
  h->uev_valid = end;
  return DSACT_OK;
}

int dsact_compute_grads(dsact_handle* h, int64_t iteration, uint32_t flags) {
  TRY(check_ready(h, true));
  HIPCHK(h, hipSetDevice(h->device));
  TRY(mark_update(h, false));
  TRY(enqueue_prologue(h, 0, iteration, 0, 1));
  TRY(enqueue_grads(h, !(flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) || (iteration % h->cfg.delay_update) == 0, false));
  return mark_update(h, true);
}

int dsact_apply_update(dsact_handle* h, int64_t iteration) {
  TRY(check_ready(h, false));
  HIPCHK(h, hipSetDevice(h->device));
  TRY(enqueue_prologue(h, 0, iteration, 1, 0));
  TRY(enqueue_adam(h));
  return (iteration % h->cfg.delay_update) == 0 ? policy_moved(h) : DSACT_OK;
}

int dsact_step(dsact_handle* h, int64_t iteration, uint32_t flags) {
  TRY(check_ready(h, true));
  HIPCHK(h, hipSetDevice(h->device));
  TRY(mark_update(h, false));
  TRY(enqueue_prologue(h, 0, iteration, 1, 1));
  // single-GPU update: Adam / Polyak are applied by the weight-gradient tiles themselves (no k_adam)
  TRY(enqueue_grads(h, !(flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) || (iteration % h->cfg.delay_update) == 0, true));
  TRY(mark_update(h, true));
  return (iteration % h->cfg.delay_update) == 0 ? policy_moved(h) : DSACT_OK;   // dsac_v2.py:320-347: the policy moves on these iterations only
}

// one replayed update (iteration and index-table row from device state). `iteration` is only used to
// decide, at enqueue/capture time, whether this is an off iteration of the delayed update.
static int enqueue_allreduce(dsact_handle* h, float* buf, size_t count, int op) {
  if (!h->comm) return fail(h, DSACT_E_STATE, "no communicator (dsact_comm_init)");
  const int rc = g_rccl.AllReduce(buf, buf, count, kNcclFloat, op, h->comm, h->stream);
  if (rc != 0) return fail(h, DSACT_E_HIP, "ncclAllReduce failed: %s", rccl_err(rc));
  return DSACT_OK;
}

// one replayed data-parallel update: gather -> local gradients -> all-reduce -> Adam/Polyak (dsac_v2.py:107-138 with the
// collective in the seam); strict mode: also the 2-float all-reduce of the std sums between the forward and the loss
static int enqueue_graph_step_dp(dsact_handle* h) {
  TRY(enqueue_gather(h, h->idx_table, h->idx_rows, 1, 0, 0));
  if (h->use_std_sums) {
    TRY(enqueue_grads(h, true, false, 1));
    TRY(enqueue_allreduce(h, h->std_sums, 2, kNcclSum));
    TRY(enqueue_grads(h, true, false, 2));
  } else {
    TRY(enqueue_grads(h, true, false));
  }
  TRY(enqueue_allreduce(h, h->grads, h->n_online + 2, kNcclAvg));
  TRY(enqueue_prologue(h, 1, 0, 1, 0));
  TRY(enqueue_adam(h));
  return policy_moved(h);
}

static int enqueue_graph_step(dsact_handle* h, long long iteration, uint32_t flags) {
  if (flags & DSACT_F_DATA_PARALLEL) return enqueue_graph_step_dp(h);
  TRY(enqueue_gather(h, h->idx_table, h->idx_rows, 1, 0, 1));
  const bool actor = !(flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) || (iteration % h->cfg.delay_update) == 0;
  return enqueue_grads(h, actor, true);
}

// captures `n` updates on the handle's stream into (*graph, *exec)
static int capture_updates(dsact_handle* h, int n, uint32_t flags, bool merged, hipGraph_t* graph, hipGraphExec_t* exec) {
  HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = DSACT_OK;
  if (!merged) {
    for (int s = 0; s < n && rc == DSACT_OK; ++s) rc = enqueue_graph_step(h, s, flags);
  } else {
    h->mirror_w0 = true;
    select_set(h, (n - 1) & 1);
    rc = enqueue_gather(h, h->idx_table, h->idx_rows, 1, 0, 1, /*bookkeeping=*/0);
    for (int s = 0; s < n && rc == DSACT_OK; ++s) {
      const int set = (n - 1 - s) & 1;
      RideArgs ride;
      memset(&ride, 0, sizeof(ride));
      select_set(h, set ^ 1);   // destination of the riding gather
      ride.g = gather_args(h, h->idx_table, h->idx_rows, 1, 0, 1);
      ride.g.lookahead = 1;
      ride.n_gather = s + 1 < n ? ride.g.n_gather_blocks : 0;
      ride.bookkeeping = 1;
      select_set(h, set);
      const bool actor = !(flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) || (s % h->cfg.delay_update) == 0;
      if (flags & DSACT_F_DATA_PARALLEL) {
        // local gradients (the next update's gather and this one's bookkeeping ride in the loss launch as on the single-GPU
        // path) -> all-reduce -> streaming Adam/Polyak -> packed weight copies rebuilt for the next forward
        rc = enqueue_grads(h, true, false, 0, &ride);
        if (rc == DSACT_OK) rc = enqueue_allreduce(h, h->grads, h->n_online + 2, kNcclAvg);
        if (rc == DSACT_OK) rc = h->env_no_adam_pack ? enqueue_adam(h) : enqueue_adam_pack(h);
        if (rc == DSACT_OK && h->env_no_adam_pack) rc = enqueue_pack(h, true);
      } else {
        rc = enqueue_grads(h, actor, true, 0, &ride);
      }
    }
    select_set(h, 0);
    h->mirror_w0 = false;
  }
  hipError_t e = hipStreamEndCapture(h->stream, graph);
  if (rc != DSACT_OK) return rc;
  if (e != hipSuccess) return fail(h, DSACT_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  HIPCHK(h, hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0));
  return DSACT_OK;
}

// `n` updates starting at an iteration with it % delay_update == phase as the PIPELINED launch sequence:
//   [gather minibatch 0 (+ repack)] [gather minibatch 1]
//   per update s:  k_chain_fwdp (own minibatch; + the policy units of minibatch s + 1 when update s leaves the policy alone)
//                  k_chain_bwd_q (+ riders: gather of minibatch s + 2, bookkeeping)   k_chain_bwd_pi (+ all dW/Adam tiles)
// Every precomputed item is produced and consumed inside one sequence: its first update never relies on one, its last
// never produces one. Same kernels behind the forward, same arithmetic per row: bit-identical to eager updates.
// The forward launches read their unit tables from `dev_args` (n PipeFwd, filled here BEFORE anything is enqueued).
struct PipePlan {
  std::vector<PipeFwd> host; std::vector<char> pre, dop;
  std::vector<char> defer;        // update s leaves the policy alone and has a successor: its policy backward rides in launch s + 1
  std::vector<BwdPiArgs> bp;      // [s]: the deferred policy backward of update s - 1 (valid when defer[s - 1])
  std::vector<int> bp_rg;
  bool dp = false;                // local gradients -> all-reduce -> k_adam_pack instead of the fused optimiser
  std::vector<char> bqt;          // update s: critics' backward + their tiles + close as one launch (its bookkeeping rides in its forward)
  std::vector<char> bqp;          // update s (moves the policy): critics' + policy backward + all tiles + close as one launch
  bool skip = false;              // DSACT_F_SKIP_ACTOR_ON_OFF_ITERS: the discarded policy backward is not computed at all
  std::vector<char> leaves;       // update s leaves policy / alpha / targets alone
};
// unit tables of the n forward launches -> plan.host and (synchronous copy: call it BEFORE a stream capture begins) dev_args
static int plan_updates_pipe(dsact_handle* h, int n, int phase, PipePlan& plan, PipeFwd* dev_args, uint32_t flags = 0) {
  const int D = h->cfg.delay_update;
  auto set_of = [&](int s) { return (s - (n - 1)) & (dsact_handle::kPipeSets - 1); };
  plan.host.resize((size_t)n); plan.pre.resize((size_t)n); plan.dop.resize((size_t)n);
  plan.defer.assign((size_t)n, 0); plan.bp.resize((size_t)n); plan.bp_rg.assign((size_t)n, 2);
  // the discarded policy backward can move when it is the merged launch's (chain + its own tiles behind the arrival counter)
  const int rg_pi = h->env_chain_rg ? h->env_chain_rg : (h->B >= 8 ? h->cRG : 1);
  // (data parallel: the same -- on those updates k_adam_pack leaves the policy alone, so the policy segment of the all-reduced
  //  arena is read by nobody; the deferred tiles store nothing and that segment simply keeps its previous content)
  const bool can_defer = h->pi_merge && h->dw_chunks == 1 && !h->fat_bwd && rg_pi <= 2 && !h->env_no_pipe_defer;
  plan.dp = (flags & DSACT_F_DATA_PARALLEL) != 0;
  plan.skip = (flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) != 0 && !plan.dp;
  plan.leaves.assign((size_t)n, 0);
  plan.bqt.assign((size_t)n, 0);
  plan.bqp.assign((size_t)n, 0);
  const bool bqt = bqt_ok(h);   // (data-parallel graphs too: the tiles then store gradients and the closing block closes nothing)
  // (4-row critic slices, 8-row policy slices: the instantiation k_chain_bwd_qpt is built for)
  const bool bqp = bqt && h->nq == 2 && !h->env_no_bqp && chain_rg(h, 4, true) == 1 && rg_pi == 2;
  if (bqt) TRY(build_bqt(h));
  bool pre = false;
  int rc = DSACT_OK;
  h->mirror_w0 = true;   // (what the enqueue pass sets: the tiles' argument blocks are built here)
  for (int s = 0; s < n && rc == DSACT_OK; ++s) {
    const bool leaves_policy = ((phase + s) % D) != 0;        // this update's close does not touch policy / alpha / targets
    const bool do_pre = s + 1 < n && leaves_policy;
    plan.leaves[(size_t)s] = leaves_policy;
    const BwdPiArgs* bp = nullptr;
    if (s > 0 && plan.defer[(size_t)s - 1]) {
      apply_pipe_set(h, set_of(s - 1));   // the deferred backward works on the PREVIOUS update's minibatch
      BwdPiArgs& a = plan.bp[(size_t)s];
      // (the deferred chain shares its launch with forward chains, not with ~500 riding tiles: 4-row slices by default)
      // (fused = false: the tiles compute, apply nothing and store nothing -- and never read the step state, which the
      //  bookkeeping block of THIS forward launch may be rewriting for the update the launch belongs to)
      bwd_pi_args(h, h->dw2_off[2], h->dw2_off[2], false, a, plan.bp_rg[(size_t)s], true);
      a.finalize = 0;                     // that update was closed by its own last launch
      a.dw.store_g = 0;                   // nothing reads this gradient (fused: no optimiser step on that update either)
      bp = &a;
    }
    plan.defer[(size_t)s] = do_pre && can_defer && !plan.skip;   // (skip: there is no policy backward to move)
    plan.bqt[(size_t)s] = plan.defer[(size_t)s] && bqt;
    plan.bqp[(size_t)s] = !leaves_policy && bqp;
    rc = pipe_fwd_build(h, set_of(s), set_of(s + 1), pre, do_pre, plan.host[(size_t)s], bp, plan.bqt[(size_t)s] != 0 || plan.bqp[(size_t)s] != 0);
    plan.pre[(size_t)s] = pre; plan.dop[(size_t)s] = do_pre;
    pre = do_pre;
  }
  h->mirror_w0 = false;
  apply_pipe_set(h, 0);
  TRY(rc);
  HIPCHK(h, hipMemcpy(dev_args, plan.host.data(), (size_t)n * sizeof(PipeFwd), hipMemcpyHostToDevice));
  return DSACT_OK;
}
static int enqueue_updates_pipe(dsact_handle* h, int n, const PipePlan& plan, PipeFwd* dev_args) {
  auto set_of = [&](int s) { return (s - (n - 1)) & (dsact_handle::kPipeSets - 1); };
  const std::vector<PipeFwd>& host = plan.host;
  const std::vector<char>&pre_v = plan.pre, &do_v = plan.dop;
  int rc = DSACT_OK;
  h->mirror_w0 = true;
  {
    // the first two minibatches (+ the packed-copy refresh: anything may have written the arenas since the last replay)
    Gather2Args g;
    apply_pipe_set(h, set_of(0));
    g.a = gather_args(h, h->idx_table, h->idx_rows, 1, 0, 1);
    g.a.bookkeeping = 0;
    g.a.rp = repack_args(h, repack_blocks(h));
    g.b = g.a;
    g.b.n_gather_blocks = 0;
    if (n > 1) {
      apply_pipe_set(h, set_of(1));
      g.b = gather_args(h, h->idx_table, h->idx_rows, 1, 0, 1);
      g.b.bookkeeping = 0; g.b.lookahead = 1;
      g.b.rp = repack_args(h, 0);
    }
    rc = launch(h, "gather", k_gather2, dim3(g.a.n_gather_blocks + g.b.n_gather_blocks + g.a.rp.n_blocks), dim3(kThreads), 0, g);
  }
  for (int s = 0; s < n && rc == DSACT_OK; ++s) {
    apply_pipe_set(h, set_of(s));
    const bool carries = s > 0 && plan.defer[(size_t)s - 1];
    rc = launch_chain_fwd_pipe(h, pipe_fwd_name(pre_v[(size_t)s], do_v[(size_t)s]), host[(size_t)s], dev_args + s,
                               carries ? &plan.bp[(size_t)s] : nullptr, plan.bp_rg[(size_t)s]);
    if (rc != DSACT_OK) break;
    RideArgs ride;
    memset(&ride, 0, sizeof(ride));
    if (s + 2 < n) {
      apply_pipe_set(h, set_of(s + 2));   // destination of the riding gather
      ride.g = gather_args(h, h->idx_table, h->idx_rows, 1, 0, 1);
      ride.g.lookahead = 2;
      ride.n_gather = ride.g.n_gather_blocks;
      apply_pipe_set(h, set_of(s));
    } else {
      ride.g = gather_args(h, h->idx_table, h->idx_rows, 1, 0, 1);   // (st / hp of the bookkeeping block)
      ride.n_gather = 0;
    }
    ride.bookkeeping = (plan.bqt[(size_t)s] || plan.bqp[(size_t)s]) ? 0 : 1;   // (merged backward launches: the forward launch did the bookkeeping)
    if (plan.dp) {
      h->pipe_defer_now = plan.defer[(size_t)s] != 0;
      h->bqt_now = plan.bqt[(size_t)s] != 0;
      h->bqp_now = plan.bqp[(size_t)s] != 0;
      rc = enqueue_grads(h, true, false, 2, &ride);
      h->pipe_defer_now = false;
      h->bqt_now = false;
      h->bqp_now = false;
      if (rc == DSACT_OK) rc = enqueue_allreduce(h, h->grads, h->n_online + 2, kNcclAvg);
      if (rc == DSACT_OK) rc = h->env_no_adam_pack ? enqueue_adam(h) : enqueue_adam_pack(h);
      if (rc == DSACT_OK && h->env_no_adam_pack) rc = enqueue_pack(h, true);
      continue;
    }
    h->pipe_defer_now = plan.defer[(size_t)s] != 0;
    h->bqt_now = plan.bqt[(size_t)s] != 0;
    h->bqp_now = plan.bqp[(size_t)s] != 0;
    rc = enqueue_grads(h, !(plan.skip && plan.leaves[(size_t)s]), true, 2, &ride);
    h->pipe_defer_now = false;
    h->bqt_now = false;
    h->bqp_now = false;
  }
  apply_pipe_set(h, 0);
  h->mirror_w0 = false;
  h->n_heads_parts = h->B / 4;
  return rc;
}

static int capture_updates_pipe(dsact_handle* h, int n, int phase, hipGraph_t* graph, hipGraphExec_t* exec, PipeFwd** dev_args, uint32_t flags) {
  HIPCHK(h, hipMalloc((void**)dev_args, (size_t)n * sizeof(PipeFwd)));
  PipePlan plan;
  TRY(plan_updates_pipe(h, n, phase, plan, *dev_args, flags));
  HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  const int rc = enqueue_updates_pipe(h, n, plan, *dev_args);
  hipError_t e = hipStreamEndCapture(h->stream, graph);
  if (rc != DSACT_OK) return rc;
  if (e != hipSuccess) return fail(h, DSACT_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  HIPCHK(h, hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0));
  return DSACT_OK;
}

// would dsact_graph_build(steps_per_graph, flags) capture the pipelined graph?
static bool pipe_eligible(const dsact_handle* h, int steps_per_graph, uint32_t flags) {
  const int D = h->cfg.delay_update;
  const bool merged = !h->cnn && h->use_w1p && h->dw_chunks == 1 && !h->use_fork && !h->use_std_sums && h->alt_ws != nullptr &&
                      !h->env_no_merged_gather;
  // (data parallel too: replicas hold identical policies, which change on the same iterations)
  // (noise comes from the device either way: Philox keyed by iteration, or the uploaded noise table of dsact_run_group)
  return merged && h->chain_ok && !h->fat && h->fwd_merge && h->B % 4 == 0 && (h->rng_seed != 0 || h->noise_table_on) &&
         (!(flags & DSACT_F_DATA_PARALLEL) || h->comm != nullptr) && D >= 2 &&
         D <= dsact_handle::kPipePhases && steps_per_graph >= 2 && !h->env_no_pipe;
}

int dsact_graph_build(dsact_handle* h, int32_t steps_per_graph, uint32_t flags) {
  TRY(check_ready(h, false));
  if (steps_per_graph < 1) return fail(h, DSACT_E_INVALID, "steps_per_graph must be >= 1");
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first (dsact_upload_index_table)");
  HIPCHK(h, hipSetDevice(h->device));
  // (argument checks first: a refused build leaves the active graph and the cache as they were -- ADVICE r5)
  if ((flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) && steps_per_graph % h->cfg.delay_update)
    return fail(h, DSACT_E_INVALID, "with DSACT_F_SKIP_ACTOR_ON_OFF_ITERS steps_per_graph must be a multiple of delay_update");
  if ((flags & DSACT_F_DATA_PARALLEL) && !h->comm) return fail(h, DSACT_E_STATE, "DSACT_F_DATA_PARALLEL needs dsact_comm_init");
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->build_keeps_cache) stash_graph(h); else drop_graphs(h);
  h->want_graph_steps = 0;
  if (h->flags_dirty && h->chain_flags) {   // not inside the graph: the captured updates clear the flags themselves
    HIPCHK(h, hipMemset(h->chain_flags, 0, kChainFlags * sizeof(int)));
    h->flags_dirty = false;
  }
  // Merged gather (MLP nets, fused single-launch-chain update): one gather launch opens the graph; every update's
  // loss launch carries the bookkeeping and the NEXT update's gather into the other batch set; the per-step repack
  // of the padded first-layer copies is done by the weight-gradient tiles themselves (FusedOpt::mir_*).
  // Update s of n uses set (n-1-s)&1, so the last staged minibatch sits in set 0 like after eager updates.
  // (data parallel: only on the chain path, whose packed copies k_pack can rebuild after the streaming optimiser)
  const bool merged = !h->cnn && h->use_w1p && h->dw_chunks == 1 && !h->use_fork && !h->use_std_sums && h->alt_ws != nullptr &&
                      !h->env_no_merged_gather && (!(flags & DSACT_F_DATA_PARALLEL) || h->chain_ok);
  h->merged_graph = merged;
  // Pipelined graph (fused single-GPU update on the row-slice chains with both in-launch hand-overs available, device RNG,
  // 2 <= delay_update <= 4, at least 2 updates per graph): one graph per phase first_iteration % delay_update
  const int D = h->cfg.delay_update;
  const bool pipe = pipe_eligible(h, steps_per_graph, flags);
  const bool was_prof = h->profiling;
  h->profiling = false;
  int rc = DSACT_OK;
  if (pipe) {
    rc = alloc_pipe_sets(h);
    for (int ph = 0; ph < D && rc == DSACT_OK; ++ph) rc = capture_updates_pipe(h, steps_per_graph, ph, &h->pgraph[ph], &h->pexec[ph], &h->pargs[ph], flags);
    h->pipe_graph = rc == DSACT_OK;
    if (rc != DSACT_OK) {   // (e.g. out of memory for the extra minibatch sets): the plain graph serves the same contract
      drop_active_graph(h);   // the partially built pipelined graph only: cached graphs of other shapes stay
      (void)hipGetLastError();
      if (h->pipe_ws) apply_pipe_set(h, 0);   // (set 0 = the workspace's own buffers, recorded once the extra sets exist)
      rc = DSACT_OK;
    }
  }
  if (!h->pipe_graph) {
    rc = capture_updates(h, steps_per_graph, flags, merged, &h->graph, &h->graph_exec);
  }
  h->profiling = was_prof;
  if (rc != DSACT_OK) { if (h->build_keeps_cache) drop_active_graph(h); else drop_graphs(h); return rc; }
  h->graph_steps = steps_per_graph;
  h->graph_flags = flags;
  h->have_batch = true;
  return DSACT_OK;
}

// n_groups back-to-back replays of the captured updates
static int launch_groups(dsact_handle* h, int64_t first_iteration, int64_t n_groups) {
  h->uev_valid = false;
  int64_t i = 0;
  if (h->pipe_graph) {
    // the pipelined graphs are captured per phase of the delayed update: replay i starts at first_iteration + i * graph_steps
    const int D = h->cfg.delay_update;
    for (; i < n_groups; ++i) {
      const int ph = (int)(((first_iteration + i * h->graph_steps) % D + D) % D);
      HIPCHK(h, hipGraphLaunch(h->pexec[ph], h->stream));
    }
    return DSACT_OK;
  }
  for (; i < n_groups; ++i) HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
  return DSACT_OK;
}

static int set_device_iteration(dsact_handle* h, long long it) {
  // it_next lives at offset 0 of DevState; back-to-back graph replays leave it where the next one starts
  if (h->dev_it_next == it) return DSACT_OK;
  HIPCHK(h, hipMemcpyAsync(&h->st->it_next, &it, sizeof(long long), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->dev_it_next = it;
  return DSACT_OK;
}

int dsact_graph_run(dsact_handle* h, int64_t first_iteration, int64_t n_steps) {
  if (!h) return DSACT_E_INVALID;
  TRY(check_handoff(h));
  if (h->state_invalid)
    return fail(h, DSACT_E_STATE, "device state is invalid after a hand-over timeout: restore parameters / optimiser state, then "
                                  "call dsact_set_state or dsact_bind_arenas");
  TRY(ensure_wanted_graph(h));
  if (!have_graph(h)) return fail(h, DSACT_E_STATE, "dsact_graph_build first");
  if (n_steps % h->graph_steps) return fail(h, DSACT_E_INVALID, "n_steps must be a multiple of steps_per_graph");
  if ((h->graph_flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) && first_iteration % h->cfg.delay_update)
    return fail(h, DSACT_E_INVALID, "first_iteration must be a multiple of delay_update for a graph captured with DSACT_F_SKIP_ACTOR_ON_OFF_ITERS");
  HIPCHK(h, hipSetDevice(h->device));
  TRY(set_device_iteration(h, first_iteration));
  TRY(launch_groups(h, first_iteration, n_steps / h->graph_steps));
  h->dev_it_next = first_iteration + n_steps;
  return policy_moved(h);
}

// The reference's loop between two sampler calls (training/trainer.py:63-82 with sample_interval = n_steps: the CNN examples run 8,
// example_train/dsacv2_cnn_carracing_offasync.py:133): n_steps x { ReplayBuffer.sample_batch (replay_buffer.py:85-90) ->
// DSAC_V2.local_update (dsac_v2.py:102-105) } as ONE graph replay. While no add_batch intervenes the ring size is constant and
// only np.random.randint consumes the NumPy stream, so the caller draws the n_steps index rows up front -- the very calls,
// in the very order, the reference makes -- and hands them over here: idx [n_steps][batch]. noise (nullable): the
// reference's torch.randn draws of those updates, [n_steps][2*B*A + 2*B] = eps_new | eps_2 | z5 | z6 per update (strict RNG
// through the graph; nullptr: device Philox keyed by iteration, which needs dsact_set_device_rng). Nothing here waits for
// the device: the rows travel through pinned staging slots, the replay counters are reset by a stream-ordered kernel, and
// captured graphs are kept per (n_steps, flags, noise mode) -- the first group of a new shape pays its capture.
int dsact_run_group(dsact_handle* h, int64_t first_iteration, int32_t n_steps, const int64_t* idx, const float* noise, uint32_t flags) {
  if (!h || !idx || n_steps < 1) return DSACT_E_INVALID;
  TRY(check_ready(h, false));
  if (!h->rb_obs) return fail(h, DSACT_E_STATE, "buffer not created");
  if (h->size == 0) return fail(h, DSACT_E_STATE, "buffer empty");
  if (!noise && h->rng_seed == 0) return fail(h, DSACT_E_STATE, "no noise source: pass the noise rows or call dsact_set_device_rng");
  if ((flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) && (n_steps % h->cfg.delay_update || first_iteration % h->cfg.delay_update))
    return fail(h, DSACT_E_INVALID, "with DSACT_F_SKIP_ACTOR_ON_OFF_ITERS a group must cover whole delay_update periods");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t B = (size_t)h->B, A = (size_t)h->A;
  const size_t n_idx = (size_t)n_steps * B, nz_row = 2 * B * A + 2 * B;
  for (size_t i = 0; i < n_idx; ++i)
    if (idx[i] < 0 || idx[i] >= h->size) return fail(h, DSACT_E_INVALID, "index %lld out of range [0,%lld)", (long long)idx[i], h->size);
  // ---- tables (their addresses and row counts are baked into every captured graph)
  if (!h->idx_table || h->idx_rows < n_steps) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graphs(h);
    if (h->idx_table) hipFree(h->idx_table);
    if (h->noise_table) { hipFree(h->noise_table); h->noise_table = nullptr; }
    h->idx_table = nullptr;
    const int rows = n_steps > 16 ? n_steps : 16;
    HIPCHK(h, hipMalloc(&h->idx_table, (size_t)rows * B * sizeof(int)));
    HIPCHK(h, hipMemset(h->idx_table, 0, (size_t)rows * B * sizeof(int)));
    h->idx_rows = rows;
  }
  if (noise && !h->noise_table) {
    HIPCHK(h, hipMalloc(&h->noise_table, (size_t)h->idx_rows * nz_row * sizeof(float)));
    HIPCHK(h, hipMemset(h->noise_table, 0, (size_t)h->idx_rows * nz_row * sizeof(float)));
  }
  // ---- the graph of this shape: active, cached, or captured now
  const bool want_table = noise != nullptr;
  if (!(have_graph(h) && h->graph_steps == n_steps && h->graph_flags == flags && h->graph_noise_table == want_table)) {
    stash_graph(h);
    if (!activate_graph(h, n_steps, flags, want_table)) {
      h->build_keeps_cache = true;
      h->noise_table_on = want_table;
      const int rc = dsact_graph_build(h, n_steps, flags);
      h->noise_table_on = false;
      h->build_keeps_cache = false;
      TRY(rc);
      h->graph_noise_table = want_table;
    }
  }
  // ---- rows -> pinned slot -> tables (stream-ordered; the slot is reused four groups later)
  const size_t need = n_idx * sizeof(int) + (noise ? (size_t)n_steps * nz_row * sizeof(float) : 0);
  if (need > h->grp_pin_bytes) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    size_t cap = need < (1u << 16) ? (1u << 16) : need;
    for (int i = 0; i < 4; ++i) {
      if (h->grp_pin[i]) { hipHostFree(h->grp_pin[i]); h->grp_pin[i] = nullptr; }
      HIPCHK(h, hipHostMalloc((void**)&h->grp_pin[i], cap, hipHostMallocDefault));
      if (!h->grp_ev[i]) HIPCHK(h, hipEventCreateWithFlags(&h->grp_ev[i], hipEventDisableTiming));
    }
    h->grp_pin_bytes = cap;
  }
  const int slot = (int)(h->grp_k++ & 3);
  HIPCHK(h, hipEventSynchronize(h->grp_ev[slot]));
  int* pin_idx = (int*)h->grp_pin[slot];
  for (size_t i = 0; i < n_idx; ++i) pin_idx[i] = (int)idx[i];
  HIPCHK(h, hipMemcpyAsync(h->idx_table, pin_idx, n_idx * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (noise) {
    float* pin_nz = (float*)(h->grp_pin[slot] + n_idx * sizeof(int));
    memcpy(pin_nz, noise, (size_t)n_steps * nz_row * sizeof(float));
    HIPCHK(h, hipMemcpyAsync(h->noise_table, pin_nz, (size_t)n_steps * nz_row * sizeof(float), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(h, hipEventRecord(h->grp_ev[slot], h->stream));
  // ---- iteration of the first update, table row 0; then the replay
  hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(64), 0, h->stream, h->st, (long long)first_iteration, 0LL);
  if (hipGetLastError() != hipSuccess) return fail(h, DSACT_E_HIP, "launch k_set_counters failed");
  TRY(launch_groups(h, first_iteration, 1));
  h->dev_it_next = first_iteration + n_steps;
  h->have_batch = true;
  return policy_moved(h);
}

int dsact_dp_begin(dsact_handle* h, int64_t first_iteration) {
  if (!h) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  return set_device_iteration(h, first_iteration);
}

int dsact_dp_enqueue_grads(dsact_handle* h, uint32_t flags) {
  TRY(check_ready(h, false));
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first");
  HIPCHK(h, hipSetDevice(h->device));
  (void)flags;
  TRY(enqueue_gather(h, h->idx_table, h->idx_rows, 1, 0, 0));
  h->have_batch = true;
  return enqueue_grads(h, true, false);
}

int dsact_dp_enqueue_grads_critic(dsact_handle* h, uint32_t flags) {
  TRY(check_ready(h, false));
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first");
  if (h->use_std_sums) return fail(h, DSACT_E_STATE, "the split gradient halves are for the one-collective mode (strict mode has its own halves)");
  HIPCHK(h, hipSetDevice(h->device));
  (void)flags;
  TRY(enqueue_gather(h, h->idx_table, h->idx_rows, 1, 0, 0));
  h->have_batch = true;
  return enqueue_grads(h, true, false, 3);
}

int dsact_dp_enqueue_grads_actor(dsact_handle* h, uint32_t flags) {
  TRY(check_ready(h, true));
  HIPCHK(h, hipSetDevice(h->device));
  (void)flags;
  return enqueue_grads(h, true, false, 4);
}

int dsact_dp_set_strict(dsact_handle* h, float* std_sums_dev) {
  if (!h) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (any_graph(h)) return fail(h, DSACT_E_STATE, "data-parallel mode is baked into the captured graph");
  if (!h->std_sums_own) h->std_sums_own = h->std_sums;
  h->use_std_sums = std_sums_dev != nullptr;
  h->std_sums = std_sums_dev ? std_sums_dev : h->std_sums_own;
  return DSACT_OK;
}

int dsact_dp_enqueue_forward(dsact_handle* h, uint32_t flags) {
  TRY(check_ready(h, false));
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first");
  if (!h->use_std_sums) return fail(h, DSACT_E_STATE, "dsact_dp_set_strict first");
  HIPCHK(h, hipSetDevice(h->device));
  (void)flags;
  TRY(enqueue_gather(h, h->idx_table, h->idx_rows, 1, 0, 0));
  h->have_batch = true;
  return enqueue_grads(h, true, false, 1);
}

int dsact_dp_enqueue_backward(dsact_handle* h, uint32_t flags) {
  TRY(check_ready(h, true));
  if (!h->use_std_sums) return fail(h, DSACT_E_STATE, "dsact_dp_set_strict first");
  HIPCHK(h, hipSetDevice(h->device));
  (void)flags;
  return enqueue_grads(h, true, false, 2);
}

int dsact_dp_enqueue_apply(dsact_handle* h) {
  TRY(check_ready(h, false));
  HIPCHK(h, hipSetDevice(h->device));
  TRY(enqueue_prologue(h, 1, 0, 1, 0));
  return enqueue_adam(h);
}

int dsact_comm_unique_id(const char* rccl_path, uint8_t id[128]) {
  if (!id) return DSACT_E_INVALID;
  if (rccl_load(rccl_path)) return DSACT_E_STATE;
  NcclUid u;
  memset(&u, 0, sizeof(u));
  if (g_rccl.GetUniqueId(&u) != 0) return DSACT_E_HIP;
  memcpy(id, u.internal, 128);
  return DSACT_OK;
}

int dsact_comm_init(dsact_handle* h, int32_t rank, int32_t world, const uint8_t id[128], const char* rccl_path) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (any_graph(h)) return fail(h, DSACT_E_STATE, "the communicator is baked into the captured graph");
  if (const char* e = rccl_load(rccl_path)) return fail(h, DSACT_E_STATE, "%s", e);
  if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  NcclUid u;
  memcpy(u.internal, id, 128);
  const int rc = g_rccl.CommInitRank(&h->comm, world, u, rank);
  if (rc != 0) { h->comm = nullptr; return fail(h, DSACT_E_HIP, "ncclCommInitRank failed: %s", rccl_err(rc)); }
  h->comm_rank = rank; h->comm_world = world;
  return DSACT_OK;
}

int dsact_comm_destroy(dsact_handle* h) {
  if (!h) return DSACT_E_INVALID;
  if (h->comm) {
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
  }
  return DSACT_OK;
}

int dsact_dp_enqueue_allreduce(dsact_handle* h) {
  TRY(check_ready(h, false));
  HIPCHK(h, hipSetDevice(h->device));
  return enqueue_allreduce(h, h->grads, h->n_online + 2, kNcclAvg);
}

static int enqueue_stats(dsact_handle* h, float* dst) {
  StatsArgs a;
  a.part_loss = h->part_loss; a.n_loss = h->B; a.part_heads = h->part_heads; a.n_heads = h->chain_ok ? h->n_heads_parts : h->n_heads_wg;
  a.log_alpha = h->online + h->n_online - 1; a.st = h->st;
  a.inv_B = 1.0f / (float)h->B; a.inv_BA = 1.0f / ((float)h->B * (float)h->A);
  a.auto_alpha = h->cfg.auto_alpha; a.alpha_fixed = h->cfg.alpha_fixed; a.out = dst;
  // a gradient computed here and not yet applied (get_remote_update_info): report the mean_std its loss used
  a.ms_tail = h->have_local_tail && h->cfg.algo == 0 ? h->grads + h->n_online : nullptr;
  a.spin_timeout = nullptr;   // a timed-out hand-over fails the reading call itself (check_handoff)
  const bool was_prof = h->profiling;
  h->profiling = false;
  int rc = launch(h, "stats", k_stats, dim3(1), dim3(64), 0, a);
  h->profiling = was_prof;
  return rc;
}

int dsact_read_stats(dsact_handle* h, float out[16]) {
  if (!h || !out) return DSACT_E_INVALID;
  TRY(check_ready(h, false));
  HIPCHK(h, hipSetDevice(h->device));
  TRY(enqueue_stats(h, h->stats));
  HIPCHK(h, hipMemcpyAsync(out, h->stats, 16 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  TRY(check_handoff(h));
  out[15] = -1.0f;
  if (h->uev_valid) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->uev0, h->uev1) == hipSuccess) out[15] = ms;
  }
  return DSACT_OK;
}

int dsact_stats_snapshot(dsact_handle* h, int32_t slot) {
  if (!h || slot < 0 || slot >= DSACT_STATS_SLOTS) return DSACT_E_INVALID;
  TRY(check_ready(h, false));
  HIPCHK(h, hipSetDevice(h->device));
  return enqueue_stats(h, h->stats + 16 * (1 + slot));
}

int dsact_stats_read(dsact_handle* h, int32_t slot, float out[16]) {
  if (!h || !out || slot < 0 || slot >= DSACT_STATS_SLOTS) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(out, h->stats + 16 * (1 + slot), 16 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_handoff(h);
}

// ---- measurement -----------------------------------------------------------------------------------
int dsact_time_steps(dsact_handle* h, int64_t first_iteration, int64_t n_steps, uint32_t flags, int32_t use_graph, float* ms_total) {
  if (!h || !ms_total || n_steps < 1) return DSACT_E_INVALID;
  const long long shadow = h->dev_it_next;
  TRY(check_ready(h, false));
  if (use_graph) h->dev_it_next = shadow;   // replays keep the device iteration in step with the host's view
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first");
  if (!h->tev0) {
    HIPCHK(h, hipEventCreate(&h->tev0));
    HIPCHK(h, hipEventCreate(&h->tev1));
  }
  TRY(set_device_iteration(h, first_iteration));
  HIPCHK(h, hipEventRecord(h->tev0, h->stream));
  if (use_graph) {
    TRY(ensure_wanted_graph(h));
    if (!have_graph(h)) return fail(h, DSACT_E_STATE, "dsact_graph_build first");
    if (n_steps % h->graph_steps) return fail(h, DSACT_E_INVALID, "n_steps must be a multiple of steps_per_graph");
    if ((h->graph_flags & DSACT_F_SKIP_ACTOR_ON_OFF_ITERS) && first_iteration % h->cfg.delay_update)
      return fail(h, DSACT_E_INVALID, "first_iteration must be a multiple of delay_update");
    TRY(launch_groups(h, first_iteration, n_steps / h->graph_steps));
    h->dev_it_next = first_iteration + n_steps;
  } else {
    h->dev_it_next = -1;
    for (int64_t i = 0; i < n_steps; ++i) TRY(enqueue_graph_step(h, first_iteration + i, flags));
  }
  HIPCHK(h, hipEventRecord(h->tev1, h->stream));
  {
    // wait for the end event by polling first: a blocking wait sleeps the thread and its wake-up (tens of us) would
    // be charged to a short timed region; after ~5 ms of polling fall back to the blocking wait
    hipError_t q = hipErrorNotReady;
    const auto t_poll = std::chrono::steady_clock::now();
    while ((q = hipEventQuery(h->tev1)) == hipErrorNotReady) {
      if (std::chrono::steady_clock::now() - t_poll > std::chrono::milliseconds(5)) break;
    }
    if (q != hipSuccess) HIPCHK(h, hipEventSynchronize(h->tev1));
  }
  HIPCHK(h, hipEventElapsedTime(ms_total, h->tev0, h->tev1));
  h->have_batch = true;   // the last update's minibatch stays staged
  h->pol_epoch++;         // (measurement entry point: the acting snapshot is refreshed lazily by the next acting call)
  return check_handoff(h);
}

int dsact_time_stage(dsact_handle* h, int32_t stage, int32_t reps, float* ms_total, double* macs) {
  if (!h || !ms_total || reps < 1) return DSACT_E_INVALID;
  TRY(check_ready(h, false));
  const int L = h->L;
  if (stage < 0 || stage >= 2 * L || (stage >= L && stage - L >= (int)h->fwd2.size()))
    return fail(h, DSACT_E_INVALID, "stage must be in 0..%d (group B has %d layers)", 2 * L - 1, (int)h->fwd2.size());
  HIPCHK(h, hipSetDevice(h->device));
  const Stage& s = stage < L ? h->fwd1[stage] : h->fwd2[stage - L];
  if (macs) {
    double m = 0;
    for (int i = 0; i < s.args.n_prob; ++i) m += (double)s.args.p[i].M * s.args.p[i].N * s.args.p[i].K;
    *macs = m;
  }
  const bool was_prof = h->profiling;
  h->profiling = false;
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0));
  HIPCHK(h, hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) TRY(run_stage(h, s));  // warm-up
  HIPCHK(h, hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps; ++i) TRY(run_stage(h, s));
  HIPCHK(h, hipEventRecord(e1, h->stream));
  HIPCHK(h, hipEventSynchronize(e1));
  HIPCHK(h, hipEventElapsedTime(ms_total, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  h->profiling = was_prof;
  return DSACT_OK;
}

int dsact_profile_step(dsact_handle* h, int64_t iteration, uint32_t flags, dsact_kernel_time* out, int32_t cap, int32_t* n) {
  if (!h || !out || !n) return DSACT_E_INVALID;
  h->pol_epoch++;
  TRY(check_ready(h, false));
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first");
  HIPCHK(h, hipSetDevice(h->device));
  TRY(set_device_iteration(h, iteration));
  for (auto& r : h->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  h->prof.clear();
  h->profiling = true;
  int rc = enqueue_graph_step(h, iteration, flags);
  h->profiling = false;
  TRY(rc);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int cnt = 0;
  for (auto& r : h->prof) {
    if (cnt >= cap) break;
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, r.e0, r.e1));
    memset(&out[cnt], 0, sizeof(out[cnt]));
    strncpy(out[cnt].name, r.name.c_str(), sizeof(out[cnt].name) - 1);
    out[cnt].ms = ms;
    out[cnt].blocks = r.blocks;
    ++cnt;
  }
  *n = cnt;
  return DSACT_OK;
}

// n consecutive updates issued eagerly as the launch sequence dsact_graph_build(n_steps, flags) would capture -- the
// pipelined sequence when that is what it would capture -- with start/stop events on every dispatch
int dsact_profile_steps(dsact_handle* h, int64_t first_iteration, int32_t n_steps, uint32_t flags, dsact_kernel_time* out, int32_t cap, int32_t* n) {
  if (!h || !out || !n || n_steps < 1) return DSACT_E_INVALID;
  h->pol_epoch++;
  TRY(check_ready(h, false));
  if (!h->idx_table) return fail(h, DSACT_E_STATE, "upload an index table first");
  HIPCHK(h, hipSetDevice(h->device));
  TRY(set_device_iteration(h, first_iteration));
  h->dev_it_next = -1;
  for (auto& r : h->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  h->prof.clear();
  int rc = DSACT_OK;
  if (pipe_eligible(h, n_steps, flags)) {
    const int D = h->cfg.delay_update;
    PipeFwd* dev = nullptr;
    TRY(alloc_pipe_sets(h));
    HIPCHK(h, hipMalloc((void**)&dev, (size_t)n_steps * sizeof(PipeFwd)));
    PipePlan plan;
    rc = plan_updates_pipe(h, n_steps, (int)((first_iteration % D + D) % D), plan, dev, flags);
    if (rc == DSACT_OK) {
      h->profiling = true;
      rc = enqueue_updates_pipe(h, n_steps, plan, dev);
      h->profiling = false;
    }
    hipStreamSynchronize(h->stream);
    hipFree(dev);
  } else {
    h->profiling = true;
    for (int i = 0; i < n_steps && rc == DSACT_OK; ++i) rc = enqueue_graph_step(h, first_iteration + i, flags);
    h->profiling = false;
  }
  TRY(rc);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_batch = true;
  int cnt = 0;
  for (auto& r : h->prof) {
    if (cnt >= cap) break;
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, r.e0, r.e1));
    memset(&out[cnt], 0, sizeof(out[cnt]));
    strncpy(out[cnt].name, r.name.c_str(), sizeof(out[cnt].name) - 1);
    out[cnt].ms = ms;
    out[cnt].blocks = r.blocks;
    ++cnt;
  }
  *n = cnt;
  return check_handoff(h);
}

int dsact_chain_active(const dsact_handle* h) { return h && h->chain_ok ? 1 : 0; }

const char* dsact_debug_names(void) {
  return "X0,XP,X2,rew,done,eps_new,eps_2,z5,z6,logits_pi,logits_pit,logp_new,logp2,"
         "qout_c0,qout_c1,qout_t0,qout_t1,qout_p0,qout_p1,dout0,dout1,dout2,dout3,dout_pi,d_new_act,"
         "part_loss,part_heads,H.<chain>.<l>,G.<chain>.<l>,dZ.<chain>.<l> (chains: pi,pit,q1c,q2c,q1t,q2t,q1p,q2p)";
}

int dsact_debug_read(dsact_handle* h, const char* name, float* out, size_t cap, size_t* n) {
  if (!h || !name || !out || !n) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t B = h->B, A = h->A;
  const float* src = nullptr;
  size_t cnt = 0;
  int unpack_w = 0;
  std::string s(name);
  struct E { const char* k; const float* p; size_t c; };
  const E tab[] = {
      {"X0", h->X0, B * h->ldx}, {"XP", h->XP, B * h->ldx}, {"X2", h->X2, B * h->ldx}, {"rew", h->rew, B}, {"done", h->done, B},
      {"eps_new", h->eps_new, B * A}, {"eps_2", h->eps_2, B * A}, {"z5", h->z5, B}, {"z6", h->z6, B},
      {"logits_pi", h->logits_pi, B * 2 * A}, {"logits_pit", h->logits_pit, B * 2 * A}, {"logp_new", h->logp_new, B}, {"logp2", h->logp2, B},
      {"qout_c0", h->qout_c[0], 2 * B}, {"qout_c1", h->qout_c[1], 2 * B}, {"qout_t0", h->qout_t[0], 2 * B}, {"qout_t1", h->qout_t[1], 2 * B},
      {"qout_p0", h->qout_p[0], 2 * B}, {"qout_p1", h->qout_p[1], 2 * B},
      {"dout0", h->dout[0], 2 * B}, {"dout1", h->dout[1], 2 * B}, {"dout2", h->dout[2], 2 * B}, {"dout3", h->dout[3], 2 * B},
      {"dout_pi", h->dout_pi, B * 2 * A}, {"d_new_act", h->d_new_act, B * A},
      {"part_loss", h->part_loss, (size_t)h->B * kLossPart}, {"part_heads", h->part_heads, (size_t)h->n_heads_wg * 2},
      {"timeline", (const float*)h->timeline, (size_t)1024 * 16 * 2},
  };
  for (const E& e : tab) if (s == e.k) { src = e.p; cnt = e.c; }
  if (!src && s.size() > 4 && (s[0] == 'H' || s[0] == 'G' || s.compare(0, 2, "dZ") == 0)) {
    const size_t d1 = s.find('.'), d2 = s.rfind('.');
    if (d1 != std::string::npos && d2 != d1) {
      const std::string kind = s.substr(0, d1), chn = s.substr(d1 + 1, d2 - d1 - 1);
      const int l = atoi(s.c_str() + d2 + 1);
      int ch = -1;
      for (int c = 0; c < N_CHAIN; ++c) if (chn == kChainName[c]) ch = c;
      if (ch >= 0 && l >= 0 && l < chain_L(h, ch)) {
        const int wch = (kChainNet[ch] == N_POL || kChainNet[ch] == N_POLT) ? h->wp[l] : h->wq[l];
        cnt = B * wch;
        if (kind == "H") src = h->Hb[ch][l];
        else if (kind == "G") src = h->Gb[ch][l];
        else if (kind == "dZ" && kDzSlot[ch] >= 0) src = h->dZ[kDzSlot[ch]][l];
        unpack_w = h->chain_ok ? wch : 0;   // chain mode keeps these as transposed packs [feature][batch]
      }
    }
  }
  if (!src && h->cnn && (s.compare(0, 5, "cact.") == 0 || s.compare(0, 4, "cdy.") == 0 || s.compare(0, 6, "dfeat.") == 0)) {
    const size_t d1 = s.find('.'), d2 = s.rfind('.');
    const int st = atoi(s.c_str() + d1 + 1), j = d2 != d1 ? atoi(s.c_str() + d2 + 1) : 0;
    if (st >= 0 && st < N_STACK && j >= 0 && j < h->n_conv) {
      const ConvGeom& g = h->cg[j];
      cnt = B * g.OH * g.OW * g.Cout;
      if (s[1] == 'a') src = h->cact[st][j];
      else if (s[1] == 'd' && st < 3) src = h->cdy[st][j];
      else if (s[1] == 'f' && st < 3) { src = h->dfeat[st]; cnt = B * h->F; }
    }
  }
  if (!src) return fail(h, DSACT_E_INVALID, "unknown debug buffer '%s'", name);
  if (cnt > cap) return fail(h, DSACT_E_INVALID, "debug buffer '%s' needs %zu floats, cap %zu", name, cnt, cap);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (unpack_w) {
    std::vector<float> raw(cnt);
    HIPCHK(h, hipMemcpy(raw.data(), src, cnt * sizeof(float), hipMemcpyDeviceToHost));
    const int C = h->B / 16;
    for (int b = 0; b < h->B; ++b)
      for (int f = 0; f < unpack_w; ++f) out[(size_t)b * unpack_w + f] = raw[pk_index(f, b, C)];
  } else {
    HIPCHK(h, hipMemcpy(out, src, cnt * sizeof(float), hipMemcpyDeviceToHost));
  }
  *n = cnt;
  return DSACT_OK;
}

namespace {
__global__ void k_debug_fill(float* p, long long rows, long long ld, int col0, int ncols, float v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ncols) return;
  p[(i / ncols) * ld + col0 + (i % ncols)] = v;
}
int debug_fill(dsact_handle* h, float* p, long long rows, long long ld, int col0, int ncols, float v) {
  if (!p || rows * ncols <= 0) return DSACT_OK;
  const long long n = rows * ncols;
  hipLaunchKernelGGL(k_debug_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, p, rows, ld, col0, ncols, v);
  return hipGetLastError() == hipSuccess ? DSACT_OK : fail(h, DSACT_E_HIP, "debug fill failed");
}
}  // namespace

int dsact_debug_set(dsact_handle* h, const char* name, double value) {
  if (!h || !name) return DSACT_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (!strcmp(name, "withhold_flag")) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graphs(h);   // the switch travels in the kernel arguments
    h->debug_withhold = (int)value;   // 1: a forward producer's ready flag; 2: a policy-backward slice's arrival
    if (h->twin && h->online) TRY(build_twin_fwd(h));   // (the twin-trunk forward reads the switch from its device-memory table)
    return DSACT_OK;
  }
  if (!strcmp(name, "raise_handoff_word")) {   // tests: what a timed-out waiter writes -- bit 0: an update kernel's word, bit 1: the acting forward's
    if (!h->handoff_host) return fail(h, DSACT_E_STATE, "no hand-off words");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((int)value & 1) ((volatile int*)h->handoff_host)[0] = 1;
    if ((int)value & 2) ((volatile int*)h->handoff_host)[1] = 1;
    return DSACT_OK;
  }
  if (!strcmp(name, "host_act")) {      // 1: acting forward on the host (default), 0: the one-launch GPU forward (A/B, tests)
    h->host_act = value != 0.0;
    h->pol_epoch++;
    return DSACT_OK;
  }
  if (!strcmp(name, "policy_dirty")) {  // the caller wrote policy parameters itself (torch ops on the arena: load_state_dict, ...)
    h->pol_epoch++;
    return DSACT_OK;
  }
  if (!strcmp(name, "ack_state")) {     // the caller restored (or accepts) the device state after a hand-over timeout
    h->state_invalid = false;
    return DSACT_OK;
  }
  if (!strcmp(name, "poison_handover")) {
    // everything a merged launch hands from producers to consumers: the saved observation parts of the critics' first
    // layers and the action columns the policy heads fill (both batch sets). A consumer that does not wait for -- or
    // does not see -- its producer's stores then computes on `value` (tests pass NaN).
    const float v = (float)value;
    for (int i = 0; i < 4; ++i) TRY(debug_fill(h, h->zobs[i], 1, 0, 0, h->B * h->w[0], v));
    float* rows[4] = {h->XP, h->X2, h->alt.XP, h->alt.X2};
    for (float* r : rows) TRY(debug_fill(h, r, h->B, h->ldx, h->F, h->A, v));
    for (int i = 0; i < 2; ++i) TRY(debug_fill(h, h->dAq[i], 1, 0, 0, h->B * 32, v));
    // merged policy backward: the policy's dZ packs its weight-gradient tiles wait for
    if (h->chain_ok) for (int l = 0; l < h->L; ++l) TRY(debug_fill(h, h->dZ[kDzSlot[C_PI]][l], 1, 0, 0, h->B * h->w[l], v));
    // merged critic backward (k_chain_bwd_qt): the critics' dZ packs and dL/dout their weight-gradient tiles wait for
    if (h->chain_ok && !h->twin)
      for (int i = 0; i < h->nq; ++i) {
        for (int l = 0; l < h->L; ++l) TRY(debug_fill(h, h->dZ[kDzSlot[i == 0 ? C_Q1C : C_Q2C]][l], 1, 0, 0, h->B * h->w[l], v));
        TRY(debug_fill(h, h->doutT[i], 1, 0, 0, 32 * h->B, v));
      }
    return DSACT_OK;
  }
  if (!strcmp(name, "fwd_merge")) {   // A/B switch of the merged forward launch on a live handle (tests)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graphs(h);
    h->fwd_merge = value != 0.0 && h->chain_ok && !h->cnn && h->B <= 256 && h->B / 4 <= kChainFlagSlices && chain_rg(h, 4) == 1;
    return DSACT_OK;
  }
  if (!strcmp(name, "pi_merge")) {    // same for the merged policy-backward / policy weight-gradient launch
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graphs(h);
    h->pi_merge = value != 0.0 && h->chain_ok && !h->cnn && h->B <= 512 && !h->fat_bwd;
    if (h->chain_flags) HIPCHK(h, hipMemset(h->chain_flags, 0, kChainFlagInts * sizeof(int)));
    h->flags_dirty = false;
    return DSACT_OK;
  }
  return fail(h, DSACT_E_INVALID, "dsact_debug_set: unknown name '%s'", name);
}

int dsact_debug_get(const dsact_handle* h, const char* name, double* value) {
  if (!h || !name || !value) return DSACT_E_INVALID;
  if (!strcmp(name, "fwd_merge")) *value = h->fwd_merge ? 1.0 : 0.0;
  else if (!strcmp(name, "pi_merge")) *value = h->pi_merge ? 1.0 : 0.0;
  else if (!strcmp(name, "dcol_ident")) {   // conv layers whose data gradient is one masked product (identity col2im, enqueue_conv_backward)
    int n = 0;
    for (int j = 1; j < h->n_conv; ++j) {
      const ConvGeom& g = h->cg[j];
      if (g.OH == 1 && g.OW == 1 && g.H == g.KS && g.W == g.KS && !h->env_no_dcol_ident && g.Cin > 16) ++n;
    }
    *value = (double)n;
  }
  else if (!strcmp(name, "act_launch_us")) *value = h->act_launch_us;
  else if (!strcmp(name, "act_host")) *value = act_host_ok(h) ? 1.0 : 0.0;
  else if (!strcmp(name, "act_host_us")) *value = h->act_host_us;
  else if (!strcmp(name, "act_host_threads")) *value = (double)h->act_threads;
  else if (!strcmp(name, "act_host_pinned")) *value = h->act_pool ? (double)h->act_pool->pinned() : 0.0;
  else if (!strcmp(name, "act_host_repins")) *value = (double)h->act_repins;
  else if (!strcmp(name, "act_host_isa")) *value = (double)hostact::cpu_isa();
  else if (!strcmp(name, "act_copy_wait_us")) *value = h->act_copy_wait_us;
  else if (!strcmp(name, "act_host_calls")) *value = (double)h->act_host_calls;
  else if (!strcmp(name, "act_copies")) *value = (double)h->act_copies;
  else if (!strcmp(name, "act_wait_us")) *value = h->act_wait_us;
  else if (!strcmp(name, "fat")) *value = (h->fat ? 1.0 : 0.0) + (h->fat_bwd ? 2.0 : 0.0);
  else if (!strcmp(name, "handoff_failures")) *value = (double)h->handoff_failures;
  else if (!strcmp(name, "graph_steps")) *value = (double)h->graph_steps;
  else if (!strcmp(name, "pipe_graph")) *value = h->pipe_graph ? 1.0 : 0.0;   // the captured graphs are the pipelined ones
  else if (!strcmp(name, "state_invalid")) *value = h->state_invalid ? 1.0 : 0.0;
  else if (!strcmp(name, "act_fast")) *value = act_fast_ok(h) ? 1.0 : 0.0;     // dsact_act_sample / the one-launch acting forward serve this handle
  else if (!strcmp(name, "graph_cache")) *value = (double)h->graph_cache.size();   // inactive captured graphs kept by dsact_run_group
  else if (!strcmp(name, "graph_noise_table")) *value = h->graph_noise_table ? 1.0 : 0.0;
  else if (!strcmp(name, "twin_par")) *value = (h->twin_par ? 1.0 : 0.0) + (h->twin_merged ? 2.0 : 0.0);   // CNN nets: trunks as own workgroups (+2: one launch)
  else return DSACT_E_INVALID;
  return DSACT_OK;
}

// the single-launch acting forward (dsact_act.h). eps == nullptr: out = the 2A logits (mean | std); else out = A actions
// followed by A per-dimension log-prob terms of TanhGaussDistribution.sample() with the caller's N(0,1) draw
static int act_forward_fast(dsact_handle* h, const float* obs_host, const float* eps, float* out_host) {
  TRY(check_handoff(h));
  ActArgs a;
  a.n_layers = h->Lp + 1;
  const float* base = net_params(h, N_POL);
  int wg = 0;
  for (int l = 0; l <= h->Lp; ++l) {
    a.ly[l].W = base + h->pd.w_off[l]; a.ly[l].b = base + h->pd.b_off[l];
    a.ly[l].K = h->pd.in[l]; a.ly[l].N = h->pd.out[l]; a.ly[l].half = 0;
    if (h->pd.nblk == 2 && l > 0 && l < h->Lp) { a.ly[l].K = h->pd.in[l] / 2; a.ly[l].half = h->pd.out[l] / 2; }   // two (H x Hprev) blocks
    a.wg_begin[l] = wg;
    wg += ((l == h->Lp && eps ? h->A : h->pd.out[l]) + 3) / 4;
  }
  a.wg_begin[h->Lp + 1] = wg;
  if (h->act_call >= 0x7ffffff0) {   // the tags only have to differ from call to call: restart far from the sign bit
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemset(h->act_h, 0, (size_t)kActMaxLayers * kMaxWidth * sizeof(unsigned long long)));
    memset(h->act_out_host, 0, 64 * sizeof(unsigned long long));
    h->act_call = 0;
  }
  a.h = h->act_h; a.call = ++h->act_call;
  a.A = h->A; a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std; a.act = h->cfg.policy_act;
  a.out = h->act_out_dev; a.timeout = h->handoff_dev + 1;   // the acting forward's own hand-off word (check_handoff)
  a.sample = eps ? 1 : 0; a.act_scale = h->act_scale; a.act_center = h->act_center;
  a.out_act = h->cfg.policy_out_act; a.out_n = h->cfg.policy_std_param ? h->A : 2 * h->A;
  if (eps) memcpy(a.eps, eps, (size_t)h->A * sizeof(float));
  memcpy(a.x, obs_host, (size_t)h->O * sizeof(float));
  const auto tl = std::chrono::steady_clock::now();
  TRY(launch(h, "act_mlp", k_act_mlp, dim3(wg), dim3(256), 0, a));
  const auto t0 = std::chrono::steady_clock::now();
  h->act_launch_us = std::chrono::duration<double, std::micro>(t0 - tl).count();
  const int n_out = 2 * h->A;
  const unsigned want = (unsigned)a.call;
  unsigned polls = 0;
  for (int k = 0; k < n_out;) {   // the data is the flag: every result arrives as a (value, call) pair
    const unsigned long long pr = ((volatile unsigned long long*)h->act_out_host)[k];
    if ((unsigned)(pr >> 32) == want) {
      const unsigned bits = (unsigned)pr;
      memcpy(out_host + k, &bits, sizeof(float));
      ++k;
      continue;
    }
    if ((++polls & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
      HIPCHK(h, hipStreamSynchronize(h->stream));   // surfaces a device fault, if that is what happened
      TRY(check_handoff(h));
      return fail(h, DSACT_E_HIP, "acting forward did not complete");
    }
  }
  h->act_wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return check_handoff(h);
}
static bool act_fast_ok(const dsact_handle* h) {
  return !h->cnn && h->O <= kActMaxObs && h->Lp + 1 <= kActMaxLayers && !h->env_no_fast_act && h->A <= 32;
}

// OffSampler.sample()'s per-step device work in ONE call (training/off_sampler.py:46-54): policy(obs) on the live weights +
// TanhGaussDistribution.sample() (utils/act_distribution_cls.py:32-42) with the caller's standard-normal draw eps[A]
// (torch.randn(1, A) consumes the torch generator exactly as Normal.sample() does and mean + std * eps IS its result):
// action[A] (inside the action limits by construction of the tanh squashing; the caller clips like the reference) and
// its log-probability. MLP policies only (DSACT_E_INVALID otherwise: the caller takes dsact_policy_forward).
int dsact_act_sample(dsact_handle* h, const float* obs_host, const float* eps_host, float* action_host, float* logp_host) {
  if (!h || !obs_host || !eps_host || !action_host || !logp_host) return DSACT_E_INVALID;
  if (!h->online) return fail(h, DSACT_E_STATE, "arenas not bound");
  if (!h->limits_set) return fail(h, DSACT_E_STATE, "action limits not set (dsact_set_action_limits)");
  if (!act_fast_ok(h)) return fail(h, DSACT_E_INVALID, "dsact_act_sample serves MLP policies with obs <= %d floats", kActMaxObs);
  HIPCHK(h, hipSetDevice(h->device));
  if (act_host_ok(h)) return act_forward_host(h, obs_host, eps_host, action_host, logp_host);   // dsact_host_act.h
  float out[64];
  TRY(act_forward_fast(h, obs_host, eps_host, out));
  const int A = h->A;
  float lp = 0.0f;
  for (int d = 0; d < A; ++d) { action_host[d] = out[d]; lp += out[A + d]; }   // Independent(..., 1): sum over the action dimensions
  *logp_host = lp;
  return DSACT_OK;
}

int dsact_policy_forward(dsact_handle* h, const float* obs_host, int32_t n, float* logits_host) {
  if (!h || !obs_host || !logits_host) return DSACT_E_INVALID;
  if (n < 1 || n > kActRows) return fail(h, DSACT_E_INVALID, "n must be 1..%d", kActRows);
  if (!h->online) return fail(h, DSACT_E_STATE, "arenas not bound");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t O = h->O, ld = h->ldx;
  if (n == 1 && act_host_ok(h)) return act_forward_host(h, obs_host, nullptr, logits_host, nullptr);   // on the calling thread (dsact_host_act.h)
  if (n == 1 && act_fast_ok(h)) return act_forward_fast(h, obs_host, nullptr, logits_host);   // one launch (dsact_act.h)
  if (h->cnn) {
    // conv stack of the online policy on n images: (C,H,W) rows -> pixel-major -> conv layers -> feature rows
    if (!h->stage_img) HIPCHK(h, hipMalloc(&h->stage_img, 2 * (size_t)h->Brows * O * sizeof(float)));
    HIPCHK(h, hipMemcpyAsync(h->stage_img, obs_host, (size_t)n * O * 4, hipMemcpyHostToDevice, h->stream));
    TRY(enqueue_gather_img(h, h->stage_img, h->stage_img, h->idx_iota, 1, 0, false, h->aimg, h->aimg, n));
    const NetDesc& d = h->pd;
    for (int j = 0; j < h->n_conv; ++j) {
      const ConvGeom& g = h->cg[j];
      ConvStageArgs a;
      memset(&a, 0, sizeof(a));
      a.g = g; a.ix = conv_index(g); a.n_prob = 1;
      ConvGroup& p = a.p[0];
      p.in = j == 0 ? h->aimg : h->aact[j - 1];
      p.n_sub = 1;
      p.w[0] = net_params(h, N_POL) + d.cw_off[j]; p.bias[0] = net_params(h, N_POL) + d.cb_off[j];
      p.out[0] = h->aact[j]; p.M = n * g.OH * g.OW; p.tiles_n = tiles_of(g.Cout, TN);
      p.item_end = tiles_of(p.M, TM) * p.tiles_n;
      a.n_items = p.item_end;
      TRY(launch(h, "act_conv", k_conv_fwd, dim3(conv_fwd_grid(a.n_items)), dim3(kThreads), 0, a));
    }
    FeatArgs f;
    memset(&f, 0, sizeof(f));
    f.act[0] = h->aact[h->n_conv - 1]; f.dst0[0] = h->Xact; f.n_stack = 1; f.B = n; f.P = h->cP; f.C = h->cg[h->n_conv - 1].Cout; f.ldx = h->ldx;
    const long long ne = (long long)n * h->F;
    TRY(launch(h, "act_feat", k_feat_scatter, dim3((unsigned)((ne + kThreads - 1) / kThreads), 1), dim3(kThreads), 0, f));
  } else {
    HIPCHK(h, hipMemcpy2DAsync(h->Xact, ld * 4, obs_host, O * 4, O * 4, n, hipMemcpyHostToDevice, h->stream));
  }
  for (int l = 0; l < h->Lp; ++l) TRY(run_stage(h, h->actf[l]));
  PolicyOutArgs a;
  a.H = h->Hact[h->Lp - 1];
  a.Wout = net_params(h, N_POL) + h->pd.w_off[h->Lp];
  a.bout = net_params(h, N_POL) + h->pd.b_off[h->Lp];
  a.W = h->wp[h->Lp - 1]; a.n = n; a.A = h->A; a.lo_ls = h->cfg.min_log_std; a.hi_ls = h->cfg.max_log_std; a.out = h->act_out;
  a.out_act = h->cfg.policy_out_act; a.out_n = h->cfg.policy_std_param ? h->A : 2 * h->A;
#define CALL_POUT(N) TRY(launch(h, "policy_out", k_policy_out<N>, dim3((n + 3) / 4), dim3(kThreads), 0, a))
  NCH_DISPATCH(a.W, CALL_POUT);
  HIPCHK(h, hipMemcpyAsync(logits_host, h->act_out, (size_t)n * 2 * h->A * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DSACT_OK;
}

}  // extern "C"
