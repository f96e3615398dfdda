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

#define GO1PPO_ABI_VERSION 1
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

const char* go1ppo_version(void);

#ifdef __cplusplus
}
#endif
#endif
