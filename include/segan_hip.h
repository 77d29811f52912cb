/*
 * segan_hip.h — C ABI of libsegan_hip.so, the MI355X (gfx950) kernels behind the
 * SEGAN+/WSEGAN GAN training step.
 *
 * The reference (santi-pdp/segan_pytorch) has no FFI: every FLOP of its hot path is
 * an implicit ATen op dispatched by a torch.nn module.  Each entry point below
 * replaces one such implicit op (the reference call site it stands in for is cited
 * per function); the Python package `segan_pytorch_amd` binds them with ctypes (see
 * INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *  - all tensors are contiguous fp32, NCL ([batch, channel, time]) like torch;
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator);
 *    the library never allocates, frees or keeps a pointer past the call;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the
 *    call returns without synchronising;
 *  - return value: 0 on success, negative on error; segan_last_error() returns a
 *    thread-local message.  Nothing throws, nothing calls exit().
 *
 * Weight tensors are [m, n, K] in both directions: Conv1d.weight = [Cout, Cin, K]
 * (m = low-rate side = Cout) and ConvTranspose1d.weight = [Cin, Cout, K] (m = Cin).
 */
#ifndef SEGAN_HIP_H
#define SEGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGAN_ABI_VERSION 13

#define SEGAN_PAD_REFLECT 0
#define SEGAN_PAD_ZERO 1

#define SEGAN_ACT_NONE 0
#define SEGAN_ACT_TANH 1

/* A logical [B, C0+C1, L] activation made of up to two channel segments, with an
 * optional per-channel transform applied while it is staged into LDS:
 *     v = x * scale[c] + shift[c];   v = v > 0 ? v : v * slope[c]
 * (each vector may be NULL = identity; c indexes the concatenated channel axis).
 * This is how PReLU (modules.py:101), BatchNorm-normalise (modules.py:100), the
 * skip scale alpha and both torch.cat's (generator.py:76,205) are folded into the
 * consumer's load instead of being materialised. */
typedef struct segan_src {
  const float* p0;
  const float* p1; /* NULL when C1 == 0 */
  int32_t C0;
  int32_t C1;
  const float* scale;
  const float* shift;
  const float* slope;
} segan_src;

int segan_abi_version(void);
const char* segan_last_error(void);

/* Workgroup slots the launch planners leave free for OTHER kernels (default 0, or the environment's
 * SEGAN_RESERVED_SLOTS): the contraction kernels run persistent grids of one workgroup per resident
 * slot with equal shares of the work, so a foreign kernel that holds n slots while one is launched
 * (RCCL's channels during a data-parallel step: one 256-thread workgroup each) delays it by a
 * whole share (+25-33 %), not by n / slots.  With a reserve of n the grids are planned for n fewer
 * workgroups.  Process-wide (an atomic: safe to call while other threads launch); returns the
 * previous value.  The data-parallel path leaves it at 0 on purpose: measured with emulated 32-channel
 * collectives behind the production reducer, a static reserve costs more (+2.3 ms per
 * step while no collective is resident) than the delayed shares it avoids (+1.6 ms) — it is a knob
 * for a first multi-GPU run, not a default.  (There is no reference counterpart: the reference is
 * single-GPU, README.md:79.) */
int segan_set_reserved_slots(int n);

/* Bytes of packed-weight workspace segan_pack_weights needs for each form. */
size_t segan_packed_f_bytes(int M, int N, int S);
size_t segan_packed_t_bytes(int M, int N, int S);

/* Re-lay w[m][n][K] into the two polyphase packings the contraction kernels read
 * (either destination may be NULL).  `pad_t` is the transposed-form padding: the
 * ConvTranspose1d padding for a deconv weight (modules.py:115), 0 for a conv
 * weight (whose T form is its data gradient in padded coordinates).
 * The packed buffers are OPAQUE operands of the entry points below: consume them only through
 * segan_conv1d_fwd / segan_deconv1d_dgrad (wf) and segan_deconv1d_fwd / segan_conv1d_dgrad (wt) called
 * with the same (M, N, K, S, pad_t).  (Since ABI v12 the F packing of a K = 31 weight with an even
 * channel count N > 2 keeps, in the padding-tap row of every odd channel, the row-30 weights of its
 * even partner — the contraction kernels merge the two half-empty MFMA steps of a channel pair —
 * so wf is no longer a plain zero-padded transpose of w.) */
int segan_pack_weights(const float* w, float* wf, float* wt, int M, int N, int K, int S,
                       int pad_t, void* stream);

/* Precision of the four forward / data-gradient contractions below (`precision` argument):
 *   SEGAN_PREC_FP32   exact fp32 on v_mfma_f32_32x32x2_f32; packed weights from
 *                     segan_pack_weights (the default, and the benchmarked configuration)
 *   SEGAN_PREC_BF16   operands rounded to bf16, fp32 accumulate (BASELINE config 5)
 *   SEGAN_PREC_BF16X3 every fp32 operand split exactly into 3 bf16 planes, 6 partial products:
 *                     fp32-class accuracy on the bf16 matrix cores
 * The two bf16 modes read weights packed by segan_pack_weights_bf (planes = 1 / 3); a
 * geometry they do not cover returns -3 (unsupported) and the caller uses fp32. */
#define SEGAN_PREC_FP32 0
#define SEGAN_PREC_BF16 1
#define SEGAN_PREC_BF16X3 3
/* fp32 with BLOCKED accumulation: the contraction runs into the MFMA accumulators for 256 terms
 * at a time and the block sums are added in a second register set, so the rounding error grows
 * with sqrt(256) + sqrt(K/256) instead of sqrt(K): forward error against fp64 3e-7 instead of
 * 1.5-2.2e-6 on the deep layers (tests/diag/diag_accum.py), at two instead of three resident
 * waves per SIMD (~3 % of the contraction rate, 1.5-2.2 % of the training step).  Same packed weights as SEGAN_PREC_FP32;
 * built for stride 4, other strides run the plain kernels.  Not a mode of segan_wgrad (its
 * contractions are split into short pieces already). */
#define SEGAN_PREC_FP32_BLOCKED 4
size_t segan_packed_bf_bytes(int M, int N, int S, int tform, int planes);
int segan_pack_weights_bf(const float* w, void* out, int M, int N, int K, int S, int tform,
                          int pad_t, int planes, void* stream);

/* Scratch of the bf16 / bf16x3 forms (precision 1 / 3) of the four contraction entry points:
 * the activation operand is converted ONCE per call into bf16 planes in tile order (transform,
 * both channel segments, padding, roll and polyphase split applied), from where both MFMA operands
 * stream into LDS by LDS-DMA.  op: 0 conv forward, 1 conv data gradient, 2 deconv forward,
 * 3 deconv data gradient; N, M as in that entry point; L = length of the HIGH-rate side; pad =
 * its padL / pad argument.  Pass a buffer of at least this size as `scratch`; without one (or for
 * a geometry the bf16 kernels do not cover) the entry point returns SEGAN_EUNSUPPORTED and the
 * caller repeats the call with SEGAN_PREC_FP32 and the fp32 weight packing. */
size_t segan_bf16_scratch_bytes(int op, int B, int N, int M, int L, int K, int S, int pad, int planes);
/* Scratch of the four forward / data-gradient contractions below (`scratch`, `scratch_bytes`;
 * may be NULL / 0).  The fp32 kernels run whole rounds of equal tiles one per workgroup and cut
 * the tiles of the last, partial round along the contraction across ALL workgroups
 * (stream-K); a cut tile's pieces are written to `scratch` as accumulator slabs and summed in
 * chunk order by a second small kernel — a fixed order, so results are bit-reproducible run
 * to run (no atomics).  segan_corr_scratch_bytes() is the size that always suffices (128 MiB);
 * with less (or NULL) the launch simply stays one-tile-per-workgroup.  The scratch must not be
 * shared by launches that may run concurrently (different streams). */
size_t segan_corr_scratch_bytes(void);
/* Diagnostics (tests): how the calling thread's last fp32 forward / data-gradient contraction
 * was launched: {kernel (1 general, 2 fast), workgroups, tiles, tiles run whole, (tile, chunk)
 * units per workgroup of the stream-K part or 0, input transform mode}. */
void segan_debug_last_corr(int* out6);
/* The same for the last fp32 weight gradient: {kernel (1 general, 2 fast), tiles, contraction
 * splits, chunks per split, resident workgroups per CU assumed, hi loads per lane}. */
void segan_debug_last_wgrad(int* out6);

/* GConv1DBlock forward without norm/activation (modules.py:91-99):
 *   out[b,m,t] = bias[m] + sum_{n,k} w[m,n,k] * pad(roll(x))[b,n,S*t+k]
 * x: [B, N, L] (segan_src), out: [B, M, L/S].  mode = reflect with
 * (K/2-1, K/2) padding (stride>1) as the reference, or zero padding `padL`.
 * `roll` is the discriminator phase shift (discriminator.py:160-172; the conv sees
 * torch.roll(x, roll, 2)).  The output is the PRE-activation; its consumer applies
 * PReLU/BN through its own segan_src. */
int segan_conv1d_fwd(const segan_src* x, const void* wf, const float* bias, float* out, int B,
                     int N, int M, int L, int K, int S, int padL, int mode, int roll,
                     int precision, void* scratch, size_t scratch_bytes, void* stream);

/* Data gradient of the above (autograd of modules.py:98-99): dx[b,n,i] += over the
 * reflect-padded, rolled coordinates.  da: [B, M, L/S]; dx: [B, N, L] is fully
 * overwritten.  `halo` is scratch of B*N*(K-1) floats.  `w` (optional) is the UNPACKED
 * weight [M][N][K]: when given and N <= 2 (the first layer: 1-2 input channels) a direct
 * VALU kernel is used instead of the MFMA tile kernel and `wt` may be NULL. */
int segan_conv1d_dgrad(const float* da, const void* wt, const float* w, float* dx, float* halo,
                       int B, int N, int M, int L, int K, int S, int padL, int roll, int precision,
                       void* scratch, size_t scratch_bytes, void* stream);

/* Weight gradient shared by both layer types (W form):
 *   dw[m,n,k] += sum_{b,t} lo[b,m,t] * pad(roll(hi))[b,n,S*t+k]
 * conv:   lo = da (grad of pre-activation), hi = layer input x, reflect padding;
 * deconv: lo = layer input x (with its transform), hi = dy, zero padding.
 * Accumulates into dw, which is how torch accumulates .grad.  The contraction over (b, t) is
 * split across workgroups; by default the partial tiles are added with fp32 atomics (order
 * varies run to run, results differ in the last bits); with SEGAN_WGRAD_DETERMINISTIC in
 * `flags` they are written to `scratch` and added in split order by a second kernel
 * (bit-reproducible; fp32 only).
 * `precision`: SEGAN_PREC_*; the bf16 modes contract 8 samples per MFMA operand and return
 * -3 (SEGAN_EUNSUPPORTED) for Ls % 4 != 0 or very short rows: the caller falls back to fp32.
 * `scratch` / `scratch_bytes` (may be NULL / 0 unless deterministic): segan_wgrad_scratch_bytes
 * (...) bytes.  fp32: a lo operand with a transform or in two segments is materialised there
 * once per call (the fast kernel streams lo by LDS-DMA), followed by the partial tiles of the
 * deterministic mode.  bf16 modes: the lo operand converted ONCE into bf16 planes laid out for
 * the kernel (every column tile of dw re-reads it). */
#define SEGAN_WGRAD_DETERMINISTIC 1
size_t segan_wgrad_scratch_bytes(int B, int M, int N, int Ls, int S, int precision, int flags);
int segan_wgrad(const segan_src* lo, const segan_src* hi, float* dw, int B, int M, int N, int Ls,
                int K, int S, int padL, int mode, int roll, int precision, int flags, void* scratch,
                size_t scratch_bytes, void* stream);

/* GDeconv1DBlock forward (modules.py:135-141): ConvTranspose1d(stride S, padding
 * `pad`) trimmed to S*Ls samples, + bias, optional tanh (last generator layer).
 *   y[b,n,j] = bias[n] + sum_{m} sum_{t,k: S*t+k-pad=j} x[b,m,t] * w[m,n,k]
 * x: [B, M, Ls] as a segan_src (the skip concat and alpha scaling of
 * generator.py:64-76 are its second segment), y: [B, N, S*Ls].  `w` (optional): the
 * UNPACKED weight; with N <= 2 (the last generator layer, Cout = 1) the direct VALU
 * kernel is used and `wt` may be NULL. */
int segan_deconv1d_fwd(const segan_src* x, const void* wt, const float* w, const float* bias,
                       float* y, int B, int M, int N, int Ls, int K, int S, int pad, int act,
                       int precision, void* scratch, size_t scratch_bytes, void* stream);

/* Data gradient of the deconv: dx[b,m,t] = sum_{n,k} w[m,n,k] * dy[b,n,S*t+k-pad].
 * The M rows are split at M0 into two destinations (dx0: [B,M0,Ls], dx1:
 * [B,M-M0,Ls]); a NULL destination skips that half's tiles entirely (the z half of
 * the first decoder layer needs no gradient). */
int segan_deconv1d_dgrad(const float* dy, const void* wf, float* dx0, float* dx1, int B, int M,
                         int M0, int N, int Ls, int K, int S, int pad, int precision, void* scratch,
                         size_t scratch_bytes, void* stream);

/* ---- per-channel pointwise / reduction kernels ------------------------------------ */

/* BatchNorm1d training statistics (modules.py:10-11; torch.nn.BatchNorm1d): per
 * channel mean and biased variance of x[B,C,L]; writes scale = gamma*rstd and
 * shift = beta - mean*scale for the consumer's segan_src, saves mean/rstd for the
 * backward, and updates running_mean / running_var (momentum, unbiased variance)
 * when they are non-NULL.  `ws` is scratch of 3*C*nsplit floats (nsplit from
 * segan_bn_nsplit, which is also the split count of act_bwd / tanh_bwd). */
int segan_bn_nsplit(int B, int C, int L);
int segan_bn_stats(const float* x, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, float* mean,
                   float* rstd, float* scale, float* shift, float* ws, int B, int C, int L,
                   void* stream);
/* The same in two calls, for synchronised BatchNorm under data parallelism: `partial` leaves
 * this rank's (count, mean, M2) per channel and batch split in ws[nsplit][C][3]; the caller
 * all-gathers the ranks' ws; `final` combines nsplit_total = world*nsplit partials (Chan et
 * al.) exactly as segan_bn_stats does for one rank. */
int segan_bn_partial(const float* x, float* ws, int B, int C, int L, void* stream);
int segan_bn_final(const float* ws, int nsplit_total, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var, float* mean,
                   float* rstd, float* scale, float* shift, int C, void* stream);

/* y = tanh(x*scale[c] + shift[c]) on [B, C, L] (scale / shift may be NULL): the generator's
 * last block when it carries a BatchNorm (modules.py:135-141 with norm_type='bnorm'). */
int segan_affine_tanh(const float* x, const float* scale, const float* shift, float* y, int B,
                      int C, int L, void* stream);
/* y = x * scale[c] * mask on [B, C, L] (scale may be NULL; y may alias x): nn.Dropout on the skip
 * path (generator.py:53-54,70-71) with mask = 0 or 1/(1-p) per element, and its backward. */
int segan_scale_mask(const float* x, const float* scale, const float* mask, float* y, int B, int C,
                     int L, void* stream);

/* y = prelu(x*scale[c] + shift[c], slope[c]) materialised (used for the FC input
 * h.view(B,-1) of discriminator.py:181 and for int_act / ret_hid outputs). */
int segan_affine_prelu(const float* x, const float* scale, const float* shift, const float* slope,
                       float* y, int B, int C, int L, void* stream);

/* GSkip with merge_mode 'sum' (generator.py:64-74): out = prelu(x0, slope0) + alpha[c]*x1
 * (slope0 NULL = identity); all [B,C,L]. */
int segan_sum_skip(const float* x0, const float* slope0, const float* x1, const float* alpha,
                   float* out, int B, int C, int L, void* stream);

/* Backward through an (optional BN) + PReLU/identity + optional alpha-skip tap of a
 * pre-activation a[B,C,L]:
 *   v  = a*scale + shift (BN folded; identity when NULL)
 *   g  = dh * (v > 0 ? 1 : slope)          (+ alpha * dskip when dskip != NULL)
 *   dslope += sum dh * min(v, 0) ; dalpha += sum dskip * a ; dbias += sum da
 *   BN: dbeta += sum g ; dgamma += sum g*xhat ; da = scale*(g - dbeta/N - xhat*dgamma/N)
 * Any gradient output may be NULL.  ws: scratch of (4*nsplit + 2)*C floats. */
int segan_act_bwd(const float* a, const float* dh, const float* dskip, const float* slope,
                  const float* alpha, const float* bn_mean, const float* bn_rstd,
                  const float* bn_gamma, const float* bn_beta, float* da, float* dslope,
                  float* dalpha, float* dgamma, float* dbeta, float* dbias, float* ws, int B, int C,
                  int L, void* stream);
/* The BatchNorm branch of segan_act_bwd in two calls (synchronised BatchNorm): `reduce`
 * accumulates dslope / dgamma / dbeta from this rank's samples and leaves the per-channel
 * (sum g, sum g*xhat) in totals[C][2]; the caller all-reduces (sums) totals; `apply` writes da
 * with the global totals and the global per-channel element count, and accumulates dbias. */
int segan_act_bwd_bn_reduce(const float* a, const float* dh, const float* slope,
                            const float* bn_mean, const float* bn_rstd, const float* bn_gamma,
                            const float* bn_beta, float* dslope, float* dgamma, float* dbeta,
                            float* totals, float* ws, int B, int C, int L, void* stream);
int segan_act_bwd_bn_apply(const float* a, const float* dh, const float* slope,
                           const float* bn_mean, const float* bn_rstd, const float* bn_gamma,
                           const float* bn_beta, const float* totals, float* da, float* dbias,
                           float* ws, int B, int C, int L, double count_total, void* stream);

/* tanh backward of the generator output with the L1 term of model.py:316-319 fused:
 *   g = dy_adv (may be NULL) + l1_scale * sign(y - clean) (when clean != NULL)
 *   da = g * (1 - y*y) ; dbias += sum da.   ws: scratch of nsplit*C floats. */
int segan_tanh_bwd(const float* y, const float* dy, const float* clean, float l1_scale, float* da,
                   float* dbias, float* ws, int B, int C, int L, void* stream);

/* ---- dense layers of the discriminator head (discriminator.py:111-117) ------------ */

/* C[M,N] (+)= op(A)[M,K] * op(B)[K,N] with explicit element strides; exact fp32 on
 * MFMA.  beta0 != 0 overwrites C, otherwise accumulates.  Small outputs split the contraction
 * across workgroups: partials are added with fp32 atomics, or — SEGAN_GEMM_DETERMINISTIC in
 * `flags` — written as slabs into `scratch` (segan_gemm_scratch_bytes, 16-byte aligned) and
 * added in split order by a second kernel (bit-reproducible); deterministic without scratch keeps
 * the contraction whole in one workgroup per tile. */
#define SEGAN_GEMM_DETERMINISTIC 1
size_t segan_gemm_scratch_bytes(int M, int N, int K);
int segan_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
               float* C, int64_t ldc, int M, int N, int K, int beta0, int flags, void* scratch,
               size_t scratch_bytes, void* stream);

/* y[r,c] = prelu(x[r,c] + bias[c], slope[c]) (slope NULL = identity), rows x cols. */
int segan_bias_prelu_rows(const float* x, const float* bias, const float* slope, float* y,
                          int rows, int cols, void* stream);
/* backward: dx = dy*(v>0?1:slope), dslope[c] += sum dy*min(v,0), dbias[c] += sum dx,
 * with v = x + bias. */
int segan_bias_prelu_rows_bwd(const float* x, const float* bias, const float* slope,
                              const float* dy, float* dx, float* dslope, float* dbias, int rows,
                              int cols, void* stream);

/* ---- losses (train.py:94 nn.MSELoss; model.py:79 F.l1_loss) ------------------------ */
/* loss[0] = mean((x - target)^2) (loss may be NULL);
 * grad (may be NULL) = 2*(x-target)/n * gscale * (gout ? gout[0] : 1), gout being the
 * upstream scalar gradient as a DEVICE pointer (no host sync). */
int segan_mse_const(const float* x, float target, float* loss, float* grad, const float* gout,
                    float gscale, int n, void* stream);
/* F.binary_cross_entropy_with_logits against a constant label (WSEGAN --vanilla_gan,
 * model.py:582-583); same argument meaning as segan_mse_const. */
int segan_bce_logits_const(const float* x, float target, float* loss, float* grad,
                           const float* gout, float gscale, int n, void* stream);
/* loss[0] = mean(|x - y|)  (n may be large; ws: scratch of 1024 floats). */
int segan_l1_mean(const float* x, const float* y, float* loss, float* ws, int64_t n,
                  void* stream);
/* grad = sign(x - y) / n * gscale * (gout ? gout[0] : 1)   (torch.sign: sign(0) = 0). */
int segan_l1_bwd(const float* x, const float* y, const float* gout, float gscale, float* grad,
                 int64_t n, void* stream);

/* Data gradient of conv1d_fwd for SHORT rows (L/S in {4, 8, 16, 32, 64}; N % 4 == 0, M % 16 == 0;
 * any supported stride), as the GEMM + col2im of a transposed conv: no zero-halo columns, the
 * reflect fold and the roll are applied in the epilogue, complete dx rows are stored (no halo
 * scratch).  `wg` is the "G" packing of the weight: wg[m][n*32 + r*U + u] = w[m][n][S*u + r]
 * (U = 32/S; 0 for S*u + r >= K), segan_packed_g_bytes() bytes, written by
 * segan_pack_weights_g.  Same result as segan_conv1d_dgrad (fp32).  Returns SEGAN_EUNSUPPORTED
 * for other geometries. */
size_t segan_packed_g_bytes(int M, int N, int S);
int segan_pack_weights_g(const float* w, float* wg, int M, int N, int K, int S, void* stream);
int segan_conv1d_dgrad_short(const float* da, const float* wg, float* dx, int B, int N, int M,
                             int L, int K, int S, int padL, int roll, void* stream);

/* Global pooling over time of x[rows][L] (rows = B*C), the 'gmax' / 'gavg' discriminator heads
 * (discriminator.py:128-137,183-190).  mode 0: y[row] = max, idx[row] = the FIRST position
 * attaining it; mode 1: y[row] = mean (idx unused, may be NULL).  bwd: dx[row][t] =
 * (t == idx[row]) ? dy[row] : 0, or dy[row] / L. */
int segan_pool_time_fwd(const float* x, float* y, int* idx, int rows, int L, int mode,
                        void* stream);
int segan_pool_time_bwd(const float* dy, const int* idx, float* dx, int rows, int L, int mode,
                        void* stream);

/* F.mse_loss between two tensors (--reg_loss mse_loss, train.py:179, model.py:79): loss[0] =
 * mean((x - y)^2) (ws: 1024 floats); grad = 2*(x - y)/n * gscale * (gout ? gout[0] : 1). */
int segan_mse_mean(const float* x, const float* y, float* loss, float* ws, int64_t n,
                   void* stream);
int segan_mse_bwd(const float* x, const float* y, const float* gout, float gscale, float* grad,
                  int64_t n, void* stream);

/* ---- spectral normalisation ('snorm', modules.py:12-14, discriminator.py:118-121) -------
 * torch.nn.utils.spectral_norm with its defaults (one power iteration in training mode,
 * eps 1e-12) on a weight [A][Bd][K] viewed as a matrix with rows = dim 0 (Conv1d, Linear,
 * PReLU) or dim 1 (ConvTranspose1d):
 *   power_iteration != 0:  v <- normalize(W^T u), u <- normalize(W v)   (u, v updated in place)
 *   sigma[0] = u . (W v);  w_sn = w / sigma
 * ws: scratch of segan_snorm_ws_floats(...) floats (shared by fwd and bwd). */
size_t segan_snorm_ws_floats(int A, int Bd, int K, int dim);
int segan_snorm_fwd(const float* w, float* u, float* v, float* w_sn, float* sigma, float* ws, int A,
                    int Bd, int K, int dim, int power_iteration, float eps, void* stream);
/* dw += dw_sn/sigma - (<dw_sn, w>/sigma^2) u v^T   (u, v: the vectors the forward used) */
int segan_snorm_bwd(const float* dw_sn, const float* w, const float* u, const float* v,
                    const float* sigma, float* dw, float* ws, int A, int Bd, int K, int dim,
                    void* stream);

/* ---- STFT power loss of the WSEGAN step (model.py:640-653) ------------------------------
 * torch.stft(x, n_fft, hop_length=hop, win_length=win, normalized=True) with window=None:
 * a rectangular window zero-padded to n_fft, centre (reflect) padding of n_fft/2.  Only `win`
 * samples of a frame are non-zero, so the transform is frames[B*NF, win] x basis[win, 2*nbins]
 * (NF = 1 + T/hop, nbins = n_fft/2+1; columns [0,nbins) real, [nbins,2nbins) imaginary) run
 * through segan_gemm.  Rows of the basis and of the spectra are `pitch` =
 * segan_stft_pitch(n_fft) floats wide (2*nbins rounded up to a multiple of 4, pad columns
 * zero) so that they are 16-byte aligned for the GEMM.  Requires n_fft/2 < T. */
int segan_stft_pitch(int n_fft);
int segan_stft_basis(float* basis, int n_fft, int win, void* stream);
int segan_stft_frames(const float* x, float* frames, int B, int T, int n_fft, int hop, int win,
                      void* stream);
/* db[r][k] = 10*log10(re^2 + im^2 + eps) of S[rows][pitch]  (eps = 10e-20, model.py:646) */
int segan_powdb(const float* S, float* db, int64_t rows, int nbins, int pitch, float eps,
                void* stream);
/* dS = ddb * d(db)/d(re, im) */
int segan_powdb_bwd(const float* S, const float* ddb, float* dS, int64_t rows, int nbins, int pitch,
                    float eps, void* stream);
/* dx[B][T] = adjoint of segan_stft_frames applied to dframes[B*NF][win] (overwrites dx) */
int segan_stft_overlap_add(const float* dframes, float* dx, int B, int T, int n_fft, int hop,
                           int win, void* stream);

/* ---- input side (se_dataset.py:108-117,196-197) ------------------------------------------
 * int16 PCM slices -> (2/65535)(x-32767)+1 -> pre-emphasis y[n] = x[n] - coef*x[n-1], computed
 * in double and rounded once like numpy does (bit-exact).  pcm: [B][2][T+1] (clean row, noisy
 * row; element 0 of a row is the wav sample preceding the slice), first[B]: 1 when the slice
 * starts its wav (then y[0] = x[0]).  Outputs clean/noisy [B][T] fp32. */
int segan_pcm16_prep(const int16_t* pcm, const unsigned char* first, float* clean, float* noisy,
                     int B, int T, double coef, void* stream);

/* ---- output side and validation --------------------------------------
 * De-emphasis x[n] = coef*x[n-1] + y[n] per row of y[rows][T] (se_dataset.py:119-126: the
 * reference's per-sample python loop at the end of SEGAN.generate, model.py:154-156) as a
 * blocked scan; coef <= 0 copies.  x may alias y. */
int segan_deemphasis(const float* y, float* x, int rows, int T, double coef, void* stream);
/* Segmental SNR of utils.py:350-395 for rows of ref / deg [rows][T]: seg[rows][nframes] (nframes
 * = segan_ssnr_frames(T, srate)) holds the clamped per-frame values, out[rows][2] = (overall
 * SNR in dB, mean segmental SNR). */
int segan_ssnr_frames(int T, int srate);
int segan_ssnr(const float* ref, const float* deg, float* seg, float* out, int rows, int T,
               int srate, double eps, void* stream);

/* ---- optimizers (model.py:219-228) ---------------------------------------------------- */
/* torch.optim.RMSprop (no momentum, not centered): sq = alpha*sq + (1-alpha)*g*g;
 * p -= lr * g / (sqrt(sq) + eps), over a flat arena of n floats. */
int segan_rmsprop_step(float* p, const float* g, float* sq, float lr, float alpha, float eps,
                       int64_t n, void* stream);
/* torch.optim.Adam without weight decay/amsgrad; `step` is the 1-based step count. */
int segan_adam_step(float* p, const float* g, float* m, float* v, float lr, float beta1,
                    float beta2, float eps, int step, int64_t n, void* stream);
int segan_fill(float* p, float value, int64_t n, void* stream);
int segan_scale(float* p, float s, int64_t n, void* stream);

/* ---- data-parallel exchange (the reference has none: README.md:79) --------
 * One communicator per process (= per GPU) over RCCL, bound at run time (dlopen of librccl.so.1:
 * a host process that already carries an RCCL, e.g. PyTorch-ROCm, keeps exactly one copy).  The
 * communicator is the only library-owned resource.  Rendezvous: rank 0 calls
 * segan_comm_unique_id, ships the segan_comm_id_bytes() bytes to the other ranks over any side
 * channel, then every rank calls segan_comm_init (a collective) with its device current.
 * segan_allreduce: in-place SUM over ranks of n floats followed by * scale (1/world = the mean
 * of the per-rank gradients: D 25.8 M floats between the D backward passes and its optimizer
 * step, G 64.8 M floats between the G backward and its step — or bucket by bucket from inside
 * the backward), enqueued on `stream`, asynchronous w.r.t. the host.  segan_broadcast: rank
 * `root`'s n floats to all (initial weights).  segan_allgather: recv[world][n] <- send[n]
 * (synchronised-BatchNorm partial statistics). */
int segan_comm_id_bytes(void);
int segan_comm_unique_id(void* id_out);
int segan_comm_init(void** comm_out, int world, int rank, const void* id);
int segan_comm_destroy(void* comm);
int segan_comm_rank(void* comm);
int segan_comm_world(void* comm);
int segan_allreduce(void* comm, float* buf, size_t n, float scale, void* stream);
int segan_broadcast(void* comm, float* buf, size_t n, int root, void* stream);
int segan_allgather(void* comm, const float* send, float* recv, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGAN_HIP_H */
