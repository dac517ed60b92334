/* bnf.h -- C ABI of the MI355X-native BayesNF ensemble engine (libbnf_hip.so).
 *
 * The reference (google/bayesnf, /root/reference) has no FFI: its hot path is
 * reached through three Python calls (src/bayesnf/spatiotemporal.py:400,529,634
 * -> src/bayesnf/inference.py fit_map:376 / fit_vi:336 / predict_bnf:461) that
 * trace into one XLA program (`jax.pmap(jax.vmap(...))`, inference.py:577-619,
 * :727-745, :474-477).  This header is the boundary a maintainer would bind in
 * place of that XLA program (ctypes stub in INTEGRATION.md); each entry point
 * cites the reference code it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every pointer marked DEVICE is HBM memory owned by the caller (the Python
 *     side uses torch-ROCm tensors purely as containers); the library owns only
 *     the opaque handle and a few HIP events.
 *   - all work is enqueued on the caller's HIP stream (`stream`, a hipStream_t
 *     passed as void*; NULL = the default stream).  Nothing synchronises unless
 *     stated.
 *   - return value 0 = ok, negative = error; bnf_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.
 *   - one handle per GPU, used from one host thread at a time.
 *   - there is NO CPU fallback: every compute entry point fails with
 *     BNF_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef BNF_H_
#define BNF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNF_ABI_VERSION 6

/* limits of the static network description */
#define BNF_MAX_INPUTS   8    /* D  : time + spatial covariates              */
#define BNF_MAX_GROUPS   12   /* feature groups (inputs, fourier_d, seasonal, interactions) */
#define BNF_MAX_LAYERS   8    /* hidden layers                                */
#define BNF_MAX_FREQS    96   /* distinct seasonal frequencies                */
#define BNF_MAX_INTERACT 16

enum {
  BNF_OK = 0,
  BNF_ERR_INVALID = -1,    /* bad argument / unsupported configuration */
  BNF_ERR_NO_DEVICE = -2,  /* no usable HIP device (no CPU fallback exists) */
  BNF_ERR_HIP = -3,        /* a HIP runtime call failed */
  BNF_ERR_STATE = -4       /* call order violated (e.g. train before bind) */
};

enum { BNF_DTYPE_F32 = 0, BNF_DTYPE_BF16 = 1,   /* arithmetic of the dense contractions; accumulation is always f32 */
       /* BASELINE.json configs[4] ("fp8 MFMA dense layers"): the row-panel pipeline with (round 5) FP8 OPERAND STORAGE for
        * the weight-gradient contractions -- the activation copies H_l leave as OCP e4m3, the backward signals dZ_l as OCP
        * e5m2 over a per-member power of two, and dK_l = H_l^T dZ_l runs on the fp8 MFMA -- and (round 6, the folded forms:
        * F + 2 <= padded feature count, i.e. every BASELINE layout) the W x W FORWARD and BACKWARD-DATA contractions on the
        * block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) out of fp8 panels in LDS, weights as e4m3 x 2^5; layer 0
        * and its backward-data stay bf16 (env BNF_FP8_CONTRACT=0 at bnf_create: storage only).  Needs the row-panel pipeline
        * (depth >= 2, padded width 256 / 512 / 1024, <= 128 padded features): bnf_create refuses other shapes. */
       BNF_DTYPE_FP8 = 2,
       /* f32 storage, accumulation and epilogues like BNF_DTYPE_F32, but the contractions run on SPLIT-bf16 MFMAs: every f32
        * operand is split in registers into two bf16 pieces (16 operand bits) and hi*hi + hi*lo + lo*hi are summed by three
        * bf16 MFMAs -- products good to ~5e-6 of the largest output (exact chain: 1e-7), the three reference goldens (< 1e-4)
        * hold, 1.8x the speed of the exact f32 MFMA chain.  compute_dtype 'fp32_split' in the Python layer -- also what its
        * estimators run when no dtype is given (both f32-class engines hold SURVEY 8d's fp32 gates verbatim); an explicit
        * 'fp32' is BNF_DTYPE_F32, the exact v_mfma_f32_32x32x2_f32 arithmetic (ABI 6; ABI 5's Python layer had mapped the
        * NAME 'fp32' here). */
       BNF_DTYPE_F32S = 3 };
enum { BNF_OBS_NORMAL = 0, BNF_OBS_NB = 1, BNF_OBS_ZINB = 2 }; /* models.py:30-33 */
enum { BNF_MODE_MAP = 0, BNF_MODE_VI = 1 };       /* MLE = MAP with prior_weight 0 (spatiotemporal.py:551) */

/* feature-group kinds, order of models.py:242-247 */
enum { BNF_GROUP_INPUT = 0, BNF_GROUP_FOURIER = 1, BNF_GROUP_SEASONAL = 2, BNF_GROUP_INTERACT = 3 };

/* Static description of the network + run.  Mirrors `model_args`
 * (spatiotemporal.py:360-370) plus the arguments of ensemble_map / ensemble_vi
 * (inference.py:510-522, 626-639).  Offsets index the packed per-member
 * parameter vector whose leaf order is the reference's `params_` tuple
 * (models.py:95-103 + sorted flax leaves; bayesnf_amd/spec.py). */
typedef struct bnf_config {
  int32_t abi_version;      /* BNF_ABI_VERSION */
  int32_t device;           /* HIP device ordinal */
  int32_t dtype;            /* BNF_DTYPE_* */
  int32_t obs_model;        /* BNF_OBS_* */
  int32_t mode;             /* BNF_MODE_* */

  /* network (models.py:197-273) */
  int32_t n_inputs;         /* D */
  int32_t width;            /* W, 1..8192 (the kernels run at the next multiple of 64 on zero-padded
                               copies of the width-dependent leaves; parameters, gradients and
                               optimiser state keep the reference's shapes) */
  int32_t depth;            /* hidden layers, 1..BNF_MAX_LAYERS */
  int32_t n_features;       /* F */
  int32_t n_params;         /* P */
  int32_t n_groups;
  int32_t group_kind[BNF_MAX_GROUPS];
  int32_t group_arg[BNF_MAX_GROUPS];        /* input column for FOURIER */
  int32_t group_ncols[BNF_MAX_GROUPS];
  int32_t group_col0[BNF_MAX_GROUPS];
  int32_t group_scale_off[BNF_MAX_GROUPS];  /* feature_inv_sp_scale{i} */
  int32_t fourier_degree[BNF_MAX_INPUTS];
  float   input_scale[BNF_MAX_INPUTS];      /* float32(input_scales) */
  int32_t n_freqs;
  float   freq[BNF_MAX_FREQS];              /* make_seasonal_frequencies, float32 */
  float   harmonic[BNF_MAX_FREQS];
  int32_t n_interact;
  int32_t interact[BNF_MAX_INTERACT][2];
  int32_t off_log_noise_scale, off_shape, off_inflated;   /* params[0..2] */
  int32_t off_bias[BNF_MAX_LAYERS + 1];     /* Dense_l/bias ; [depth] = output layer */
  int32_t off_kernel[BNF_MAX_LAYERS + 1];   /* Dense_l/kernel (in,out) row-major */
  int32_t off_layer_scale[BNF_MAX_LAYERS];  /* inv_sp_layer_scale{l} */
  int32_t off_output_scale;
  int32_t off_lsa;                          /* log_scale_adjustment (D) */
  int32_t off_act_weight;                   /* logit_activation_weight */

  /* run (inference.py:510-522 / 626-639) */
  int64_t n_rows;           /* N training rows held in X / y */
  int64_t batch;            /* B rows per step (== N: full batch, no shuffling) */
  int32_t members;          /* ensemble members on THIS device */
  int64_t member_offset;    /* global id of local member 0: random streams are keyed
                               by the global id, so results do not depend on sharding */
  int32_t vi_samples;       /* S = sample_size_divergence (VI), else 1 */
  int32_t forward_only;     /* 1: handle is used for bnf_forward / quantiles only
                               (no backward buffers are carved, bnf_train refuses) */
  int32_t pipeline;         /* train-step kernels.  0 auto: one kernel per layer, with the last hidden
                               layer + output layer + likelihood + its backward in ONE kernel when
                               the width is 64/128/256 (512: bf16 only).  1: one kernel per layer and
                               every activation materialised (validation; bnf_debug_activation can
                               read all of them).
                               3: row-panel forward + backward kernel (bnf_panel.h: bf16, depth 2,
                               width 256/512, <= 128 features) -- what 0 selects where it applies */
  float   learning_rate;
  float   prior_weight;     /* 1 MAP, 0 MLE (inference.py:561-569) */
  float   kl_weight;        /* VI (inference.py:689-702) */
  uint64_t seed;
} bnf_config;

typedef struct bnf_handle bnf_handle;

/* ---- lifecycle ----------------------------------------------------------- */
int bnf_abi_version(void);
const char* bnf_last_error(void);

/* Validates cfg, selects the device, precomputes launch geometry. */
int bnf_create(const bnf_config* cfg, bnf_handle** out);
void bnf_destroy(bnf_handle* h);

/* Sizes (bytes) of the caller-provided device buffers. */
size_t bnf_workspace_bytes(const bnf_handle* h);  /* activations, gradients, packed weights */
size_t bnf_state_bytes(const bnf_handle* h);      /* optimiser state: MAP 2*E*P f32 (m,v); VI 4*E*P */
size_t bnf_param_bytes(const bnf_handle* h);      /* MAP E*P f32 ; VI 2*E*P (mu then rho) */
/* Device memory the engine has allocated ITSELF so far (everything else is the caller's): today only the work
 * buffers of bnf_row_keys -- 4 x members x n_rows x 4 bytes + (members + 1) x 4 + the radix sort's scratch --
 * allocated at the first epoch that draws its shuffles on the device, all or nothing (a failed allocation
 * returns BNF_ERR_HIP from bnf_train and leaves nothing behind), freed by bnf_destroy.  0 before that. */
size_t bnf_owned_bytes(const bnf_handle* h);

/* Attach device buffers.  X: (N, D) f32 row-major, y: (N,) f32 (replicated
 * closed-over constants of ensemble_map, inference.py:553-554).  Zeroes the
 * workspace and precomputes the seasonal feature table (models.py:62-76; the
 * features of the raw time index are data-constant).  */
int bnf_bind(bnf_handle* h, void* params /*DEVICE*/, void* opt_state /*DEVICE*/,
             void* workspace /*DEVICE*/, const float* X /*DEVICE*/,
             const float* y /*DEVICE*/, void* stream);

/* ---- training -------------------------------------------------------------- */
/* Initial values (inference.py:399-427 MAP/MLE ; :203-231 VI): Dense kernels ~
 * TruncatedNormal(0,1,[-2,2]) from the counter RNG keyed by (seed, global member,
 * index); log_noise_scale = log_noise_init (MAP: log(nanstd(y)/2), VI: 0);
 * everything else 0; VI rho = softplus^-1(0.3).  Resets optimiser state + step. */
int bnf_init_params(bnf_handle* h, float log_noise_init);

/* The reference's OWN initial parameters for the user's seed, drawn on the device from their keys (what fit() uses;
 * bnf_init_params above draws same-law values from the engine's generator).  MAP / MLE (inference.py:399-427) and
 * the VI surrogate means (:203-231) are one JointDistribution sample per member: one key per leaf
 * (bayesnf_amd/jaxseed.py map_leaf_keys / vi_mean_leaf_keys restate the split chain on the host), Dense kernels
 * ~ TruncatedNormal(0, 1, -2, 2) through jax.random.truncated_normal -- threefry2x32 bits, uniform on
 * [erf(-sqrt2), erf(sqrt2)), sqrt2 erfinv, clip -- every other leaf 0, log_noise_scale = log_noise_init;
 * VI: rho = softplus^-1(0.3).  Resets the optimiser state and the step counter like bnf_init_params.
 *   leaf_keys    DEVICE uint32 (members, n_leaves, 2)
 *   leaf_offsets HOST int32 (n_leaves + 1): offsets of the packed leaves, [0] = 0, [n_leaves] = P; n_leaves <= 64 */
int bnf_init_params_keys(bnf_handle* h, const uint32_t* leaf_keys, const int32_t* leaf_offsets, int32_t n_leaves,
                         float log_noise_init);

/* ensemble_map._run / tfp.vi.fit_surrogate_posterior_stateless (inference.py:
 * 577-619 / 727-738): `num_epochs` x (N // B) Adam steps for every local member,
 * enqueued back to back with no host round trip.  losses: DEVICE (members,
 * num_epochs) f32, receives the per-epoch mean of the pre-update step losses
 * (inference.py:614); for VI one "epoch" is one step and the value is already
 * multiplied by kl_weight (inference.py:758).  `epoch0` = index of the first
 * epoch (continuation of an earlier call; keys the shuffles). */
int bnf_train(bnf_handle* h, int64_t epoch0, int64_t num_epochs, float* losses /*DEVICE*/);

/* VI only: draw `n_draws` parameter vectors per member from the fitted
 * surrogate (inference.py:741-745).  out: DEVICE (n_draws, members, P) f32. */
int bnf_vi_posterior_draws(bnf_handle* h, int32_t n_draws, float* out /*DEVICE*/);

/* The reference's OWN minibatch shuffles for MAP / MLE (optional; without it every member shuffles each
 * epoch with the engine's keyed Feistel bijection -- same law, other numbers).  ensemble_map
 * (inference.py:571-575, 593-597, 622) permutes the data set per member and per epoch with
 * jax.random.permutation(permute_seed, N), permute_seed from the member's key chain; bayesnf_amd/jaxseed.py
 * restates that chain on the host.
 *   tables DEVICE int32 (n_epochs, members, (N / batch) * batch): row ids of the epochs
 *          [epoch0, epoch0 + n_epochs) of bnf_train's epoch counter, step s of an epoch reading
 *          columns [s * batch, (s + 1) * batch); values in [0, N).  Caller-owned, must stay alive while
 *          those epochs are enqueued AND executing; epochs outside the range use the engine's shuffle.
 * NULL restores the engine's shuffle.  Full-batch handles ignore it (no shuffle there). */
int bnf_row_tables(bnf_handle* h, const int32_t* tables, int64_t epoch0, int64_t n_epochs);

/* The same shuffles drawn ON THE DEVICE from their keys (what fit() uses: no per-epoch host work, no row-id upload --
 * at 10^7 rows a host-side table is 40 MB per member and epoch).  jax.random.permutation(permute_seed, N) is
 * `rounds` = ceil(3 ln N / ln(2^32 - 1)) times { permute_seed, sub = split(permute_seed); stable sort of the current
 * order by random_bits(sub, (N,)) } (jax/_src/random.py _shuffle); the caller supplies the sub keys
 * (bayesnf_amd/jaxseed.py map_shuffle_subkeys), the engine draws the bits (threefry2x32) and sorts (radix, stable)
 * when an epoch starts, on the handle's stream.
 *   keys DEVICE uint32 (n_epochs, members, rounds, 2) for the epochs [epoch0, epoch0 + n_epochs); caller-owned,
 *        alive while those epochs are enqueued and executing.  BNF_ERR_INVALID if `rounds` is not jax's count for N.
 * Work buffers (4 x members x N x 4 bytes + sort scratch) are allocated by the engine at the first such epoch
 * (the only allocation the engine makes itself: bnf_owned_bytes reports it) and freed by bnf_destroy.
 * BNF_ERR_INVALID when members x N exceeds 2^31 - 1 (32-bit sort offsets): use bnf_row_tables or the engine's
 * index-free shuffle for such fits.  A table from bnf_row_tables covering the same epoch wins.  NULL switches it off.
 * VI handles (ABI 6): ensemble_vi draws ONE batch per optimisation step, shared by every member --
 * `jax.random.permutation(seed, arange(N))[:batch_size]` with the seed tfp hands `target_log_prob_fn` at that step
 * (inference.py:704-709) -- so `keys` is uint32 (n_steps, 1, rounds, 2) for the steps [epoch0, epoch0 + n_steps)
 * (bayesnf_amd/jaxseed.py vi_batch_subkeys), the permutation is drawn when the step is enqueued and its first `batch`
 * entries are the rows of every member; work buffers 4 x N x 4 bytes + sort scratch. */
int bnf_row_keys(bnf_handle* h, const uint32_t* keys, int64_t epoch0, int64_t n_epochs, int32_t rounds);

/* The reference's OWN random stream for the VI noise (optional; without it the noise comes from the
 * engine's counter-based generator -- same law, other numbers).  tfp.vi.fit_surrogate_posterior_stateless
 * (inference.py:727-738) draws, at every step, `vi_samples` joint samples of the surrogate, each leaf
 * with jax.random.normal(key, (members, *leaf shape)); ensemble_vi's posterior draws (:741-753) likewise.
 * The keys are pure functions of the user's seed (threefry split / fold_in chains: bayesnf_amd/jaxseed.py
 * computes them on the host once per fit); the normals themselves are generated on the device.
 *   step_keys DEVICE uint32 (n_steps, vi_samples, n_leaves, 2): row i serves the i-th bnf_train step
 *             counted from this call; training beyond the table fails with BNF_ERR_STATE
 *   draw_keys DEVICE uint32 (n_draws, n_leaves, 2) for bnf_vi_posterior_draws
 *   leaf_offsets HOST int32 (n_leaves + 1): offsets of the parameter leaves, in the reference's order
 * Both tables are caller-owned and must stay alive; NULL, NULL restores the engine's generator.
 * Install them AFTER bnf_init_params / bnf_bind: both reset the step counter and drop installed tables. */
int bnf_vi_noise_keys(bnf_handle* h, const uint32_t* step_keys, int64_t n_steps, const uint32_t* draw_keys,
                      int64_t n_draws, const int32_t* leaf_offsets, int32_t n_leaves);

/* ---- prediction ------------------------------------------------------------ */
/* forecast_inner over all members (inference.py:103-126,129-200): theta is
 * DEVICE (n_members, P) f32 (any count; processed in chunks), Xnew DEVICE
 * (n_rows, D) f32.  loc: DEVICE (n_members, n_rows) f32 = network output;
 * aux: DEVICE (n_members, 3) f32 = {noise scale 0.01+exp(lns), softplus(shape),
 * sigmoid(inflated_loc_probs)}.  Does not touch the training state. */
int bnf_forward(bnf_handle* h, const float* theta, int64_t n_members,
                const float* Xnew, int64_t n_rows, float* loc, float* aux);

/* Quantiles of the equal-weight mixture of Normals over members, per row
 * (inference.py:42-100).  means DEVICE (n_members, n_rows), scales DEVICE
 * (n_members,), q HOST (n_q,), out DEVICE (n_q, n_rows).  approximate=0:
 * Chandrupatla root of mean_e Phi((x-mu_e)/sigma_e) - q on
 * [min mu - 5 max sigma, max mu + 5 max sigma], value tolerance 1e-5, <= 60
 * iterations; approximate=1: moment-matched Normal. */
int bnf_normal_mixture_quantiles(bnf_handle* h, const float* means, const float* scales,
                                 int64_t n_members, int64_t n_rows, const float* q,
                                 int32_t n_q, int32_t approximate, float* out);

/* Count observation models (handle created with BNF_OBS_NB / BNF_OBS_ZINB):
 * per-member forecast means and quantiles of the equal-weight mixture over members
 * (inference.py:271-333 and :497-502; TFP NegativeBinomial / ZeroInflatedNegativeBinomial
 * mean, stddev, cdf).  loc DEVICE (n_members, n_rows) and aux DEVICE (n_members, 3) as
 * written by bnf_forward; means DEVICE (n_members, n_rows) out; q HOST (n_q,) in (0,1);
 * out DEVICE (n_q, n_rows) = ceil of the Chandrupatla root of mean_e cdf_e(x) - q on
 * [0, max mean + 1.1 rsqrt(1-q) max stddev] (value tolerance 1e-5, <= 60 iterations),
 * 0 where mean_e pmf_e(0) > q.  n_q = 0: means only. */
int bnf_count_mixture_quantiles(bnf_handle* h, const float* loc, const float* aux,
                                int64_t n_members, int64_t n_rows, const float* q,
                                int32_t n_q, float* means, float* out);

/* ---- introspection used by tests and bench.py ------------------------------ */
/* One forward+backward of every local member on batch `step` of `epoch` WITHOUT
 * the optimiser update: grads DEVICE (members*S, P) f32 receives d(step loss)/d
 * theta of the likelihood part + prior part exactly as Adam would consume it;
 * loss DEVICE (members*S,) f32 the step loss.  theta_eff may be NULL (use the
 * bound params; VI: the z samples of that step). */
int bnf_debug_loss_and_grad(bnf_handle* h, int64_t epoch, int64_t step,
                            float* grads, float* loss);
/* Row ids the engine uses for (epoch, step): out DEVICE (members, B) int32
 * (MAP: per-member shuffle, inference.py:593-597; VI: one shared batch,
 * :704-709; full batch: 0..N-1). */
int bnf_debug_row_index(bnf_handle* h, int64_t epoch, int64_t step, int32_t* out);
/* VI noise of a step: out DEVICE (members, S, P) f32. */
int bnf_debug_vi_eps(bnf_handle* h, int64_t step, float* out);
/* Verification hook: from now on every VI step takes its reparameterisation noise from `eps`
 * (DEVICE, (members, S, P) f32 standard normals, caller-owned, read at each step; NULL = the
 * engine's own counter-based generator again).  It lets a test feed the noise of
 * tfp.vi.fit_surrogate_posterior_stateless (inference.py:727-738) and compare with the reference's
 * golden predictions element-wise; not meant for production sizes. */
int bnf_debug_vi_noise(bnf_handle* h, const float* eps);
/* Copy of an internal activation buffer, as f32: what = 0 features H0 (E',B,F),
 * 1+l hidden output H_{l+1} (E',B,W) for l < depth-1 (the last hidden output is
 * never stored), 100+l pre-activation A_l (E',B,W), 200 network output (E',B),
 * 300+l dZ_l (E',B,W), 400 dH0 (E',B,F). Valid after bnf_debug_loss_and_grad. */
int bnf_debug_activation(bnf_handle* h, int32_t what, float* out);
/* Raw C = A * Bt^T of the dense-contraction core (A (M,K), Bt (N,K), K a multiple
 * of 64, dtype = handle dtype, inputs given as f32 and converted) -> C (M,N) f32. */
int bnf_debug_gemm_nt(bnf_handle* h, const float* A, const float* Bt, int32_t M,
                      int32_t N, int32_t K, float* C);

/* Raw C = A^T * B of the weight-gradient core on row-major operands (A (R,M), B (R,N),
 * R a multiple of 64, M and N multiples of 8; inputs f32, converted to the handle dtype)
 * -> C (M,N) f32. */
int bnf_debug_gemm_tn(bnf_handle* h, const float* A, const float* B, int32_t R, int32_t M,
                      int32_t N, float* C);

/* Test hook: fills all 160 KiB of LDS of every CU with `pattern`-derived garbage (NaN bit patterns
 * for pattern = 0x7fc00000).  LDS is not cleared between kernels: a kernel that reads LDS it has not
 * written sees whatever the previous kernel left there, so parity tests run with poisoned LDS. */
int bnf_debug_poison_lds(bnf_handle* h, uint32_t pattern);

/* Per-kernel HIP-event timing on the handle's stream.  kernel = "*" brackets
 * every launch with an event pair, a kernel name (as reported by
 * bnf_profile_read, e.g. "gemm_fwd") only that kernel, NULL switches it off.
 * Enable, run bnf_train, then read (read synchronises on the recorded events).
 * names/avg_ms/calls: arrays of length *n (in: capacity, out: used). */
int bnf_profile_enable(bnf_handle* h, const char* kernel);
int bnf_profile_read(bnf_handle* h, int32_t* n, const char** names, double* avg_ms,
                     int64_t* calls);
/* Algorithmic FLOPs of one launch of kernel `name` for the bound configuration
 * (DESIGN.md section 4), 0 for non-contraction kernels. */
double bnf_kernel_flops(const bnf_handle* h, const char* name);

/* ---- the job's one data collective: the posterior gather -----------------------------------
 * Replaces what `jax.pmap` does implicitly when the reference's fit / predict return
 * (/root/reference/src/bayesnf/inference.py:452 `np.array(params_i)`, :486-492 the per-device
 * predictive means concatenated on the default device): every rank contributes its members'
 * device-resident parameters / predictive means and receives everyone's.  One RCCL all-gather
 * over xGMI; librccl.so is dlopen'ed on first use (the engine itself has no link dependency).
 *   id       BNF_COMM_ID_BYTES bytes made by rank 0 (bnf_comm_unique_id) and handed to every
 *            rank by the host (any side channel: the Python layer broadcasts it)
 *   send     DEVICE, bytes_per_rank bytes;  recv DEVICE, world * bytes_per_rank bytes, rank-major
 *   stream   hipStream_t the collective is enqueued on (no host synchronisation) */
#define BNF_COMM_ID_BYTES 128
typedef struct bnf_comm bnf_comm;
int bnf_comm_available(void);   /* 0 when librccl.so and its symbols resolve (local, cheap: no id, no listener) */
int bnf_comm_unique_id(void* id);
int bnf_comm_create(const void* id, int32_t world, int32_t rank, int32_t device, bnf_comm** out);
int bnf_allgather(bnf_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream);
/* ONE process driving n devices -- the reference's own shape (`jax.pmap` over `jax.local_devices()`,
 * inference.py:573-579,445): `bnf_comm_create_local` makes one communicator per listed device in one call
 * (ncclCommInitAll; out[n]; a device listed twice is refused), `bnf_allgather_group` enqueues every local rank's
 * all-gather in one group (send[i] / recv[i] / streams[i] on comms[i]'s device; recv[i] receives all n blocks;
 * `comms` must be the WHOLE set of one bnf_comm_create_local call in rank order -- checked, BNF_ERR_INVALID
 * otherwise: a partial set would leave ncclGroupEnd waiting; streams == NULL or a NULL entry = the default stream).
 * `bnf_comm_available` already requires the group symbols, so callers agree on the full capability up front. */
int bnf_comm_create_local(int32_t n, const int32_t* devices, bnf_comm** out);
int bnf_allgather_group(int32_t n, bnf_comm* const* comms, const void* const* send, void* const* recv,
                        size_t bytes_per_rank, void* const* streams);
void bnf_comm_destroy(bnf_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* BNF_H_ */
