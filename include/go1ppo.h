/* go1ppo.h — C-ABI of the fused PPO-update kernels (libgo1ppo.so, gfx950).
 *
 * The reference runs the PPO update of go1_gym_learn/ppo_cse/ppo.py:99-205 as ~350 autograd kernels per mini-batch.
 * Here the GEMMs stay in hipBLASLt (through torch.mm on pre-allocated buffers); everything between them is one of
 * the kernels below.  Activations and their gradients are bf16 row-major matrices with an explicit leading
 * dimension (in elements), parameters' gradients are fp32 and are accumulated with atomics (the caller zeroes them).
 * All functions enqueue on `stream` (a hipStream_t; capturable into a HIP graph) and return 0, or a negative
 * error code for invalid arguments; they never synchronise.
 *
 * bf16 matrices are passed as `void*`; "rows" is the mini-batch size, "cols" a multiple of 8, 16-byte aligned.
 */
#ifndef GO1PPO_H
#define GO1PPO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO1PPO_ABI_VERSION 3
#define GO1PPO_MAX_ACTIONS 32

/* y[r, c] = elu(y[r, c] + sum_i lat[r, i] * wz[c, i] (c < lat_cols)) in place.  lat/wz may be NULL (plain ELU).
 * Replaces `actor_body[0]`'s latent columns + nn.ELU of the reference ActorCritic (actor_critic.py:52-77). */
int go1ppo_elu_fwd(void* y, int64_t rows, int cols, int ld, const void* lat, int lat_ld, int npv, const void* wz,
                   int wz_ld, int lat_cols, void* stream);

/* out[r, c] = d[r, c] * elu'(.) evaluated from the activation output h (h > 0 ? 1 : h + 1); out may alias d (in
 * place) or be a column block of a wider matrix; h == NULL: identity.
 * bias_grad[c] += sum_r out[r, c] when bias_grad != NULL. */
int go1ppo_elu_bwd(const void* d, int ld_d, const void* h, int ld_h, int64_t rows, int cols, float* bias_grad, void* out,
                   int ld_out, void* stream);

typedef struct {
  /* network outputs of this mini-batch (bf16, heads padded to `head_ld` columns) */
  const void* mean;          /* [rows][head_ld], first num_actions valid */
  const void* value;         /* [rows][head_ld], column 0 valid */
  const float* std;          /* [num_actions] */
  int32_t head_ld, num_actions;
  int64_t rows;
  /* rollout storage, flattened (T*N, .) fp32, gathered through idx (int64 [rows]) */
  const int64_t* idx;
  const float* actions;      /* [.][num_actions] */
  const float* old_mu;       /* [.][num_actions] */
  const float* old_sigma;    /* [.][num_actions] */
  const float* old_logp;     /* [.] */
  const float* advantages;   /* [.] */
  const float* returns;      /* [.] */
  const float* old_values;   /* [.] */
  /* hyper-parameters (PPO_Args) */
  float clip_param, value_loss_coef, entropy_coef;
  int32_t use_clipped_value_loss;
  /* outputs */
  void* d_mean;              /* [rows][head_ld] bf16: d loss / d mean   (columns >= num_actions untouched) */
  void* d_value;             /* [rows][head_ld] bf16: d loss / d value  (columns >= 1 untouched) */
  float* d_std;              /* [num_actions]  += */
  float* d_mean_bias;        /* [num_actions]  += column sums of d_mean */
  float* d_value_bias;       /* [1]            += */
  float* kl;                 /* [1] += mean KL(old || new) */
  float* value_loss;         /* [1] += */
  float* surrogate_loss;     /* [1] += */
} Go1PpoLossArgs;

/* Surrogate + clipped value + entropy loss of ppo.py:112-150, forward and analytic gradient in one pass. */
int go1ppo_loss(const Go1PpoLossArgs* args, void* stream);

/* Adaptation-module regression (ppo.py:163-190): mse(pred[:num_train, sel], target[:num_train, sel]) and its
 * gradient; the remaining rows only feed the test loss.  pred: bf16 [rows][pred_ld]; target: fp32 storage
 * [.][npv] gathered through idx; selective != 0 -> column 0 only. */
int go1ppo_mse(const void* pred, int pred_ld, const float* target, int npv, const int64_t* idx, int64_t rows,
               int64_t num_train, int selective, void* d_pred, float* d_pred_bias, float* train_loss,
               float* test_loss, void* stream);

/* dW[n, k] (fp32, ld ldw) += sum_r dz[r, n] * h[r, k]; bf16 MFMA, split over row chunks with atomic accumulation.
 * bias_grad[n] += sum_r dz[r, n] when bias_grad != NULL (the layer's bias gradient rides along: the dz tile is in
 * LDS anyway).  n and k multiples of 64. */
int go1ppo_wgrad(const void* dz, int ld_dz, const void* h, int ld_h, int64_t rows, int n, int k, float* dW, int ldw,
                 float* bias_grad, void* stream);

/* ---- fused MLP tail, forward: for every net, out_l = act(in_l W_l^T + b_l) through up to 4 layers with the
 * activations kept on chip (actor_critic.py:52-77: the nn.Sequential bodies behind their first layer).
 * W_l: bf16 [n_out][k_in] row-major, bias bf16 [n_out]; k_in multiple of 32 and <= 512, n_out multiple of 16 and
 * <= 512; out (bf16 [rows][ld_out]) may be NULL for activations nobody reads; elu != 0 applies ELU. */
#define GO1PPO_TAIL_MAX_LAYERS 4
#define GO1PPO_TAIL_MAX_NETS 3
typedef struct {
  const void* W;
  const void* bias;
  void* out;
  int32_t n_out, k_in, ld_out, elu;
} Go1PpoTailLayer;
typedef struct {
  const void* in;            /* bf16 [rows][ld_in], first layer's k_in columns used */
  int64_t rows;
  int32_t ld_in, num_layers;
  /* input activation applied while the rows are staged (inference engines: the first layer's ELU pass disappears):
   * elu_in != 0: x <- elu(x + sum_q latent[row][q] * wz[col][q]) (latent == NULL: no latent term), exactly what
   * go1ppo_elu_fwd writes for these columns; the pre-activations in memory stay untouched */
  int32_t elu_in, npv;
  const void* latent;        /* bf16 [rows][lat_ld] or NULL */
  const void* wz;            /* bf16 [k_in][wz_ld] */
  int32_t lat_ld, wz_ld;
  Go1PpoTailLayer layer[GO1PPO_TAIL_MAX_LAYERS];
} Go1PpoTailNet;
typedef struct {
  int32_t num_nets, _pad;
  Go1PpoTailNet net[GO1PPO_TAIL_MAX_NETS];
} Go1PpoTailArgs;
int go1ppo_tail_fwd(const Go1PpoTailArgs* args, void* stream);

/* One weight-gradient problem of a batched launch (same contract as go1ppo_wgrad).  The caller fills the first ten
 * fields on the host, go1ppo_wgrad_plan() fills chunk_rows / wg_offset and returns the number of workgroups; the
 * table is then copied to device memory once and every backward pass is a single go1ppo_wgrad_batched() launch. */
typedef struct {
  const void* dz;
  const void* h;
  float* dW;
  float* bias_grad;          /* may be NULL */
  int64_t rows;
  int32_t ld_dz, ld_h, n, k, ldw;
  int32_t chunk_rows, wg_offset;      /* filled by go1ppo_wgrad_plan */
  /* structural zeros: dW[r][c] for r < zero_n, zero_k0 <= c < zero_k1 receives nothing (zero_n = 0: no mask).  The augmented
   * first-layer rows carry the privileged observations for the critic only (actor_critic.py:44-47, 58-61): the adaptation
   * module's and the actor's rows of W1 have no weight — hence no gradient — on those columns. */
  int32_t zero_n, zero_k0, zero_k1;
  /* go1ppo_wgrad_tn_batched only (NULL for the 64-tile kernels).  partials != NULL: the workgroup of row chunk s STORES its partial
   * tile into the slab partials[s * partial_stride + r * ldw + c] (fp32, plain stores; structural zeros are skipped, so a slab
   * buffer allocated as zeros keeps them) instead of adding it to dW with fp32 atomics; the slabs — ceil(rows / chunk_rows) of
   * them — are summed in a fixed order by go1ppo_grad_reduce or by the optimiser's norm pass (Go1PpoGradPiece).  Measured on
   * MI355X: the atomics of the 9-problem PPO-pass set cost 15.6 of its 64.5 us (profiles/r04_wgrad_atomics_probe.txt). */
  float* partials;
  int64_t partial_stride;
} Go1PpoWgradProblem;

int go1ppo_wgrad_plan(Go1PpoWgradProblem* host_problems, int count);
int go1ppo_wgrad_batched(const Go1PpoWgradProblem* device_problems, int count, int total_workgroups, void* stream);

/* ---- rollout glue (PPO.act / process_env_step / RolloutStorage of the reference, ppo.py:60-97, rollout_storage.py:54-84) ---- */

/* a = mean + std * noise; log N(a; mean, std) summed over actions; policy outputs written into the storage slot:
 * actions, mu, sigma [rows][num_actions], values, logp [rows].  mean/value: bf16 [rows][head_ld] head outputs. */
int go1ppo_act(const void* mean, const void* value, int head_ld, const float* std, int num_actions, int64_t rows,
               const float* noise, float* actions, float* mu, float* sigma, float* values, float* logp, void* stream);

/* rewards_out = rewards + gamma * values * time_outs (time_outs may be NULL), dones_out = dones,
 * env_bins_out = (float)env_bins (env_bins may be NULL). */
int go1ppo_store_step(const float* rewards, const uint8_t* dones, const uint8_t* time_outs, const int32_t* env_bins,
                      const float* values, float gamma, int64_t n, float* rewards_out, uint8_t* dones_out,
                      float* env_bins_out, void* stream);

/* GAE(lambda) backward scan over [T][N] buffers; returns and un-normalised advantages (= returns - values);
 * stats[0] += sum(adv), stats[1] += sum(adv^2) in double. */
int go1ppo_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, int T, int64_t N,
               float gamma, float lam, float* returns, float* advantages, double* stats, void* stream);

/* adv = (adv - mean) / (std + 1e-8) with the unbiased std of stats = [sum, sum of squares, count]. */
int go1ppo_normalize(float* adv, int64_t n, const double* stats, void* stream);

/* ---- observation ring (replaces the (T, N, H * num_obs) history block of rollout_storage.py:36-38) ----
 * ring: bf16 [T + H - 1][N][no] (no even); the history window of rollout step s is rows s .. s + H - 1 (oldest first).
 * Augmented GEMM rows (Kp bf16, Kp even, Kp >= H * no + 1 + npv): [window | 1 | privileged obs | 0 ...]. */

/* rows 0 .. H - 1 of the ring <- the environments' fp32 history windows hist[n * ld_hist + j * no + c] (first step of a
 * rollout: from there on only the newest observation is appended, history_wrapper.py:23's sliding window). */
int go1ppo_ring_snapshot(const float* hist, int64_t ld_hist, int64_t N, int H, int no, void* ring, void* stream);

/* One rollout step: ring_dst (row s + H - 1 of the ring, or NULL when the snapshot already wrote it) <- bf16(obs);
 * X (N x Kp) <- augmented rows of the windows starting at ring_window (= row s), the newest entry taken from `obs`;
 * obs_store / priv_store (fp32, or NULL) <- copies of obs (N x no) and priv (N x npv). */
int go1ppo_ring_step(const float* obs, const float* priv, void* ring_dst, const void* ring_window, int64_t N, int H, int no,
                     int npv, int Kp, void* X, float* obs_store, float* priv_store, void* stream);

/* X (rows x Kp) <- augmented rows of the storage entries idx[i] = s * N + n; priv_store: fp32 [T * N][npv]. */
int go1ppo_ring_gather(const void* ring, const float* priv_store, const int64_t* idx, int64_t rows, int64_t N, int H, int no,
                       int npv, int Kp, void* X, void* stream);

/* ---- optimiser step (ppo.py:126-160: adaptive-KL learning rate, clip_grad_norm_, Adam) ---- */

/* number of floats `partial` must hold */
int go1ppo_opt_partials(void);

/* partial[b] = block b's sum of (g * gscale)^2 over [0, n) (partial == NULL: skipped); step[0] += 1;
 * kl != NULL: lr[0] <- KL-adaptive schedule of ppo.py:126-136 on kl[0] * kl_scale. */
int go1ppo_opt_prestep(const float* g, int64_t n, float gscale, float* partial, float* step, float* lr, const float* kl,
                       float kl_scale, float desired_kl, float lr_min, float lr_max, void* stream);

/* The flat gradient as a list of pieces, ascending and disjoint (elements outside every piece are plain):
 *   kind 0: g[begin, begin + count) is final as it stands;
 *   kind 1: g[begin + i] <- sum over b < slabs of ((const float*)src)[b * stride + i]   (Go1PpoWgradProblem.partials);
 *   kind 2: the same with bf16 slabs (the row-chunk partial products of go1ppo_sum_partials' caller).
 * zero_rows > 0: the elements of the first zero_rows rows (rows of `cols` elements, counted from `begin`) on the columns
 * [zero_c0, zero_c1) are written as exact zeros.  begin, count multiples of 8 for kinds 1 / 2 (stride too); src 16-byte aligned. */
typedef struct {
  int64_t begin, count;
  const void* src;
  int64_t stride;
  int32_t kind, slabs, cols, zero_rows, zero_c0, zero_c1;
} Go1PpoGradPiece;
/* materialise the slab pieces (kinds 1, 2) into g: what a consumer other than go1ppo_opt_prestep_pieces needs — the data-parallel
 * all-reduce of the gradient, a torch optimiser. `device_pieces`: the table in device memory. */
int go1ppo_grad_reduce(float* g, const Go1PpoGradPiece* device_pieces, int num_pieces, void* stream);
/* go1ppo_opt_prestep over [0, n) with the slab pieces summed on the way: g receives the sums (so go1ppo_opt_adam reads a plain
 * gradient) and the norm partials are taken from them — one pass instead of a reduction launch + the norm pass.  Elements of
 * [0, n) outside every piece count as plain.  partial must not be NULL. */
int go1ppo_opt_prestep_pieces(float* g, int64_t n, const Go1PpoGradPiece* device_pieces, int num_pieces, float gscale, float* partial,
                              float* step, float* lr, const float* kl, float kl_scale, float desired_kl, float lr_min, float lr_max,
                              void* stream);

/* Adam (no weight decay, no amsgrad) on the elements [start0, start0+count0) U [start1, start1+count1) of the flat
 * parameter p with gradient g * gscale * clip, clip = min(1, max_norm / (sqrt(sum partial) + 1e-6)) (partial == NULL:
 * no clipping).  Refreshes the compute copies of the touched elements: body[i] (bf16) for i < n_body, tail[i - n_body]
 * (fp32) for i - n_body < n_tail behind it (elements beyond have no compute copy).  count0 + count1 == 0 is allowed (only
 * zero_slot is cleared).  step / lr are the device scalars go1ppo_opt_prestep maintains.
 * zero_grad != 0: every visited g[i] is cleared after it has been read (the following backward pass accumulates into a
 * clean gradient without a fill pass of its own); zero_slot (or NULL): one more float cleared — the KL accumulator that
 * rides in the gradient's padding. */
/* extras (may be NULL): transposes — dst[c * rows + r] (bf16) <- the refreshed weight at [start + r * cols + c]: K-contiguous
 * copies of the layers whose input gradient runs on go1ppo_gemm_nt, kept current by the optimiser step itself instead of by a
 * transpose-copy launch per backward pass. */
#define GO1PPO_ADAM_MAX_TRANSPOSES 2
typedef struct {
  int32_t num_transposes, _pad;
  struct { int64_t start; int32_t rows, cols; void* dst; } transpose[GO1PPO_ADAM_MAX_TRANSPOSES];
} Go1PpoAdamExtras;
int go1ppo_opt_adam(float* p, float* g, float* m, float* v, int64_t start0, int64_t count0, int64_t start1,
                    int64_t count1, float gscale, const float* partial, float max_norm, const float* step, const float* lr,
                    float beta1, float beta2, float eps, void* body, int64_t n_body, float* tail, int64_t n_tail, int zero_grad,
                    float* zero_slot, const Go1PpoAdamExtras* extras, void* stream);

/* out[r][c] (fp32, rows x cols contiguous) = sum over b < count of partials[b * stride + r * cols + c] (bf16) — the row-chunk
 * partial products of the first-layer weight gradient (a manual split-K over hipBLASLt's batched GEMM) summed into the flat
 * gradient; the columns [zero_c0, zero_c1) of the first zero_rows rows are written as exact zeros (the structural zeros of
 * Go1PpoWgradProblem).  cols, stride multiples of 8; 16-byte aligned. */
int go1ppo_sum_partials(const void* partials, int count, int64_t stride, int64_t rows, int cols, float* out, int zero_rows,
                        int zero_c0, int zero_c1, void* stream);

/* ---- MLP-layer GEMM with fused epilogue (replaces torch.addmm + F.elu / the ELU-backward map around it) ---- */

/* C (M x N, bf16, row stride ldc) = epilogue(A (M x K, bf16, lda) * B^T (B: N x K, bf16, ldb) + bias (fp32[N] or NULL)).
 * epilogue 0: nothing more; 1: ELU on the columns [elu_c0, elu_c1); 2: multiplied element-wise by elu'(H[m][n])
 * (H: the layer's post-ELU activations, bf16, ldh) — the input gradient of a hidden layer, B being the transposed
 * weight.  K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0, ldc/ldh % 4 == 0, A/B 16-byte and C/H 8-byte aligned; bias 16-byte
 * (fp32) or 8-byte (bf16) aligned. */
typedef struct Go1PpoGemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* H;
  int32_t M, N, K, lda, ldb, ldc, ldh, epilogue, elu_c0, elu_c1;
  int32_t elu_skip_c0, elu_skip_c1;      /* epilogue 1: columns [elu_skip_c0, elu_skip_c1) stay pre-activations (0, 0: none) */
  int32_t bias_bf16, _pad;               /* bias_bf16 != 0: `bias` points to bf16 values (the compute copy of the parameters) */
} Go1PpoGemmArgs;
int go1ppo_gemm_nt(const Go1PpoGemmArgs* args, void* stream);
/* two independent problems with equal tile grids (ceil(M / 128) * ceil(N / 128)) and the same epilogue in one launch */
int go1ppo_gemm_nt_pair(const Go1PpoGemmArgs* a, const Go1PpoGemmArgs* b, void* stream);

/* the same weight gradients on 128 x 128 tiles (LDS-DMA staging, hardware transpose reads): what the first-layer
 * gradients (n = 256 .. 1280, k = 2112) and the batched tails use.  Same problem table as go1ppo_wgrad_plan /
 * go1ppo_wgrad_batched, with rows % 64 == 0 and n, k % 8 == 0 instead of n, k % 64 == 0. */
int go1ppo_wgrad_tn_plan(Go1PpoWgradProblem* problems, int count);
int go1ppo_wgrad_tn_batched(const Go1PpoWgradProblem* device_problems, int count, int total_workgroups, void* stream);

/* ---- the 256 -> 128 -> 64 end of an MLP with both weight matrices resident in LDS (csrc/go1ppo_mlp.h) ----
 * Replaces, per net, F.elu + nn.Linear + F.elu + nn.Linear of actor_critic.py:51-92 (forward) and their autograd
 * backward up to the gradient w.r.t. the 256-wide pre-activation.  Up to GO1PPO_MLP2_MAX_NETS nets per launch. */
#define GO1PPO_MLP2_MAX_NETS 3
typedef struct {
  void* x;                     /* bf16 [rows][ld_x], 256 columns: pre-activations, replaced IN PLACE by elu(x) when
                                  elu_input != 0; already activated inputs otherwise */
  const void* W2; const void* b2;   /* bf16 [128][256], bf16 [128] */
  const void* W3; const void* b3;   /* bf16 [64][128], bf16 [64] */
  void* z2;                    /* out: bf16 [rows][ld_z2], elu(h0 W2^T + b2), 128 columns */
  void* out;                   /* out: bf16 [rows][ld_out], z2 W3^T + b3, 64 columns */
  int64_t rows;
  int32_t ld_x, ld_z2, ld_out, elu_input;
} Go1PpoMlp2Fwd;
int go1ppo_mlp2_fwd(const Go1PpoMlp2Fwd* nets, int count, void* stream);

typedef struct {
  const void* d_out;           /* bf16 [rows][ld_dout], 64 columns: gradient w.r.t. the head output */
  const void* z2; const void* h;    /* the forward pass's z2 (128 columns) and activated input elu(x) (256 columns) */
  const void* W2; const void* W3;
  void* d_z2;                  /* out: bf16 [rows][ld_dz2] = (d_out W3) * elu'(z2) */
  void* d_x;                   /* out: bf16 [rows][ld_dx] = (d_z2 W2) * elu'(h): gradient w.r.t. the pre-activation x */
  int64_t rows;
  int32_t ld_dout, ld_z2, ld_h, ld_dz2, ld_dx, _pad;
} Go1PpoMlp2Bwd;
int go1ppo_mlp2_bwd(const Go1PpoMlp2Bwd* nets, int count, void* stream);

const char* go1ppo_version(void);

#ifdef __cplusplus
}
#endif
#endif
