// segan_pointwise.hip — HBM-bound per-channel kernels of the SEGAN GAN step (gfx950):
// BatchNorm statistics, the backward of (BN +) PReLU / alpha-skip / tanh with their
// per-channel parameter gradients, the dense-head bias/PReLU, losses and the fused
// optimizers.  All reductions are two-stage (per-workgroup partials, then one
// finalising thread per channel) so results do not depend on scheduling order.
#include "segan_common.h"

#define PW_THREADS 256

// number of batch splits per channel for the [B,C,L] reductions
static int pw_nsplit(int B, int C, int L) {
  int ns = ceil_div(2048, C);
  if (ns > B) ns = B;
  // keep at least ~2048 elements per workgroup
  long per = (long)B * L / ns;
  while (ns > 1 && per < 2048) {
    ns = (ns + 1) / 2;
    per = (long)B * L / ns;
  }
  return ns < 1 ? 1 : ns;
}

extern "C" int segan_bn_nsplit(int B, int C, int L) {
  if (B <= 0 || C <= 0 || L <= 0) return 0;
  return pw_nsplit(B, C, L);
}

// block-wide sum of up to 4 values; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = warp_sum(v[k]);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) sm[wave * NV + k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float s = 0.f;
      for (int w = 0; w < PW_THREADS / 64; ++w) s += sm[w * NV + k];
      v[k] = s;
    }
  }
  __syncthreads();
}

struct ChanRange {
  int c, b_beg, b_end;
};
__device__ __forceinline__ ChanRange chan_range(int B, int nsplit) {
  ChanRange r;
  r.c = blockIdx.x;
  const int per = (B + nsplit - 1) / nsplit;
  r.b_beg = blockIdx.y * per;
  r.b_end = min(B, r.b_beg + per);
  return r;
}

// ---------------------------------------------------------------------------------
// BatchNorm statistics
// ---------------------------------------------------------------------------------
// Chan et al. combination of two (count, mean, M2) partials
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb,
                                             float qb) {
  const float tot = n + nb;
  if (tot <= 0.0f) return;
  const float d = mb - mean;
  const float f = nb / tot;
  mean = fmaf(d, f, mean);
  m2 = m2 + qb + d * d * n * f;
  n = tot;
}

// ONE pass over the slice (round 2 read it twice: mean, then squared deviations).  Every thread
// keeps shifted sums s1 = sum(x - K), s2 = sum((x - K)^2) with K = the FIRST element it visits —
// a sample of the very distribution, so |mean - K| is of the order of the spread and
// M2 = s2 - s1^2/n loses at most a bit or two (a shift far from the data, e.g. zero or a running
// mean, is what makes one-pass variances cancel) — then the threads' (count, mean, M2) are merged
// with Chan's formula in a fixed tree: lanes by shuffles, waves through LDS.
__global__ void bn_partial_kernel(const float* __restrict__ x, float* __restrict__ ws, int B, int C,
                                  int L, int nsplit) {
  __shared__ float sm[3 * (PW_THREADS / 64)];
  const ChanRange cr = chan_range(B, nsplit);
  const int nb = cr.b_end - cr.b_beg;
  // slice = rows (b, cr.c, :) for b in [b_beg, b_end): `tr` threads walk a row (float4 when
  // 4 | L), PW_THREADS/tr rows at a time; 32-bit index math only
  const float* base = x + ((size_t)cr.b_beg * C + cr.c) * L;
  const size_t sstride = (size_t)C * L;
  const bool vec = (L & 3) == 0 && ((uintptr_t)x & 15) == 0;
  const int Lv = vec ? L >> 2 : L;
  int tr = PW_THREADS;
  while (tr > Lv && tr > 1) tr >>= 1;
  const int rp = PW_THREADS / tr;
  const int t0 = threadIdx.x % tr, r0 = threadIdx.x / tr;
  float K = 0.0f, s1 = 0.0f, s2 = 0.0f, n = 0.0f;
  if (r0 < nb && t0 < Lv) K = base[(size_t)r0 * sstride + (vec ? 4 * t0 : t0)];
  for (int b = r0; b < nb; b += rp) {
    const float* row = base + b * sstride;
    if (vec) {
      for (int t = t0; t < Lv; t += tr) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(row + 4 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = u[e] - K;
          s1 += d;
          s2 = fmaf(d, d, s2);
        }
        n += 4.0f;
      }
    } else {
      for (int t = t0; t < Lv; t += tr) {
        const float d = row[t] - K;
        s1 += d;
        s2 = fmaf(d, d, s2);
        n += 1.0f;
      }
    }
  }
  float mean = 0.0f, m2 = 0.0f;
  if (n > 0.0f) {
    const float r = s1 / n;
    mean = K + r;
    m2 = fmaxf(s2 - s1 * r, 0.0f);
  }
  // lanes: butterfly-free fixed tree (lane i takes lane i + o)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb2 = __shfl_down(n, o, 64), mb2 = __shfl_down(mean, o, 64), qb2 = __shfl_down(m2, o, 64);
    chan_combine(n, mean, m2, nb2, mb2, qb2);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sm[3 * wave] = n; sm[3 * wave + 1] = mean; sm[3 * wave + 2] = m2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < PW_THREADS / 64; ++w) chan_combine(n, mean, m2, sm[3 * w], sm[3 * w + 1], sm[3 * w + 2]);
    float* w = ws + ((size_t)blockIdx.y * C + cr.c) * 3;
    w[0] = n;
    w[1] = mean;
    w[2] = m2;
  }
}

// BatchNorm as the per-channel affine map the consumers apply: y = fmaf(x, scale, shift).  One
// definition for the forward (bn_final_kernel) and the backward (act_bwd_kernel).
__device__ __forceinline__ void bn_affine(float gamma, float beta, float mean, float rstd,
                                          float& scale, float& shift) {
  scale = gamma * rstd;
  shift = fmaf(-mean, scale, beta);
}

__global__ void bn_final_kernel(const float* __restrict__ ws, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var,
                                float* mean_o, float* rstd_o, float* scale_o, float* shift_o, int C,
                                int nsplit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // Chan et al. parallel combination, in double
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int s = 0; s < nsplit; ++s) {
    const float* w = ws + ((size_t)s * C + c) * 3;
    const double nb = w[0], mb = w[1], qb = w[2];
    if (nb <= 0.0) continue;
    const double tot = n + nb;
    const double d = mb - mean;
    mean += d * nb / tot;
    m2 += qb + d * d * n * nb / tot;
    n = tot;
  }
  const double var = n > 0 ? m2 / n : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.0f;
  const float bt = beta ? beta[c] : 0.0f;
  float sc, sh;
  bn_affine(g, bt, (float)mean, rstd, sc, sh);
  if (mean_o) mean_o[c] = (float)mean;
  if (rstd_o) rstd_o[c] = rstd;
  if (scale_o) scale_o[c] = sc;
  if (shift_o) shift_o[c] = sh;
  if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unb = n > 1 ? m2 / (n - 1.0) : var;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

extern "C" int segan_bn_stats(const float* x, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* mean,
                              float* rstd, float* scale, float* shift, float* ws, int B, int C,
                              int L, void* stream) {
  SEGAN_REQUIRE(x && ws, "bn_stats: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "bn_stats: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int ns = pw_nsplit(B, C, L);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(C, ns), dim3(PW_THREADS), 0, st, x, ws, B, C, L, ns);
  hipLaunchKernelGGL(bn_final_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, ws, gamma, beta, eps,
                     momentum, running_mean, running_var, mean, rstd, scale, shift, C, ns);
  return segan_check_launch("bn_stats");
}

// segan_bn_stats in two calls for synchronised BatchNorm: `partial` leaves this rank's
// (count, mean, M2) per channel and batch split in ws[nsplit][C][3]; after an all-gather of the
// ranks' ws, `final` combines nsplit_total = world * nsplit partials exactly like segan_bn_stats.
extern "C" int segan_bn_partial(const float* x, float* ws, int B, int C, int L, void* stream) {
  SEGAN_REQUIRE(x && ws, "bn_partial: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "bn_partial: bad sizes");
  const int ns = pw_nsplit(B, C, L);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(C, ns), dim3(PW_THREADS), 0, (hipStream_t)stream, x, ws,
                     B, C, L, ns);
  return segan_check_launch("bn_partial");
}

extern "C" int segan_bn_final(const float* ws, int nsplit_total, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* mean, float* rstd, float* scale,
                              float* shift, int C, void* stream) {
  SEGAN_REQUIRE(ws && nsplit_total > 0 && C > 0, "bn_final: bad arguments");
  hipLaunchKernelGGL(bn_final_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, ws,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift,
                     C, nsplit_total);
  return segan_check_launch("bn_final");
}

// ---------------------------------------------------------------------------------
// y = prelu(x*scale + shift, slope)
// ---------------------------------------------------------------------------------
__global__ void affine_prelu_kernel(const float* __restrict__ x, const float* scale,
                                    const float* shift, const float* slope, float* __restrict__ y,
                                    size_t total, int C, int L) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    float v = x[i];
    v = fmaf(v, scale ? scale[c] : 1.0f, shift ? shift[c] : 0.0f);
    if (slope) v = v > 0.f ? v : v * slope[c];
    y[i] = v;
  }
}

extern "C" int segan_affine_prelu(const float* x, const float* scale, const float* shift,
                                  const float* slope, float* y, int B, int C, int L, void* stream) {
  SEGAN_REQUIRE(x && y, "affine_prelu: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "affine_prelu: bad sizes");
  const size_t total = (size_t)B * C * L;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(affine_prelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale,
                     shift, slope, y, total, C, L);
  return segan_check_launch("affine_prelu");
}

// y = tanh(x*scale + shift): the last generator block when it carries a BatchNorm
// (GDeconv1DBlock with norm_type='bnorm' and act='Tanh', modules.py:135-141)
__global__ void affine_tanh_kernel(const float* __restrict__ x, const float* scale,
                                   const float* shift, float* __restrict__ y, size_t total, int C,
                                   int L) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    y[i] = tanhf(fmaf(x[i], scale ? scale[c] : 1.0f, shift ? shift[c] : 0.0f));
  }
}

extern "C" int segan_affine_tanh(const float* x, const float* scale, const float* shift, float* y,
                                 int B, int C, int L, void* stream) {
  SEGAN_REQUIRE(x && y, "affine_tanh: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "affine_tanh: bad sizes");
  const size_t total = (size_t)B * C * L;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(affine_tanh_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale,
                     shift, y, total, C, L);
  return segan_check_launch("affine_tanh");
}

// y = x * scale[c] * mask: nn.Dropout on the skip path (generator.py:53-54,70-71) — mask holds
// 0 or 1/(1-p) per element — with the alpha skip scale folded in (scale may be NULL)
__global__ void scale_mask_kernel(const float* __restrict__ x, const float* scale,
                                  const float* __restrict__ mask, float* __restrict__ y,
                                  size_t total, int C, int L) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    y[i] = x[i] * (scale ? scale[c] : 1.0f) * mask[i];
  }
}

extern "C" int segan_scale_mask(const float* x, const float* scale, const float* mask, float* y,
                                int B, int C, int L, void* stream) {
  SEGAN_REQUIRE(x && mask && y, "scale_mask: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "scale_mask: bad sizes");
  const size_t total = (size_t)B * C * L;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(scale_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale,
                     mask, y, total, C, L);
  return segan_check_launch("scale_mask");
}

// out = prelu(x0, slope0) + alpha * x1   (GSkip with merge_mode 'sum', generator.py:64-74)
__global__ void sum_skip_kernel(const float* __restrict__ x0, const float* slope0,
                                const float* __restrict__ x1, const float* alpha,
                                float* __restrict__ out, size_t total, int C, int L) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    float v = x0[i];
    if (slope0) v = v > 0.f ? v : v * slope0[c];
    out[i] = fmaf(alpha[c], x1[i], v);
  }
}

extern "C" int segan_sum_skip(const float* x0, const float* slope0, const float* x1,
                              const float* alpha, float* out, int B, int C, int L, void* stream) {
  SEGAN_REQUIRE(x0 && x1 && alpha && out, "sum_skip: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "sum_skip: bad sizes");
  const size_t total = (size_t)B * C * L;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(sum_skip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x0, slope0,
                     x1, alpha, out, total, C, L);
  return segan_check_launch("sum_skip");
}

// ---------------------------------------------------------------------------------
// backward of (BN +) PReLU + alpha-skip tap
// ---------------------------------------------------------------------------------
struct ActBwdArgs {
  const float* a;
  const float* dh;
  const float* dskip;
  const float* slope;
  const float* alpha;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  float* da;
  float* ws;      // [nsplit][C][4] partials, then [C][2] totals of (dbeta, dgamma)
  int B, C, L, nsplit;
  float inv_count;   // 1 / (elements per channel the BatchNorm statistics were taken over)
};

// PHASE 0: no BN (single pass).  PHASE 1: BN reductions.  PHASE 2: BN apply.
template <int PHASE>
__global__ void act_bwd_kernel(const ActBwdArgs p) {
  __shared__ float sm[16];
  const ChanRange cr = chan_range(p.B, p.nsplit);
  const int c = cr.c;
  const int nb = cr.b_end - cr.b_beg;
  const float sl = p.slope ? p.slope[c] : 1.0f;
  const float al = p.alpha ? p.alpha[c] : 0.0f;
  float mu = 0.f, rs = 1.f, ga = 1.f, be = 0.f, dbeta_m = 0.f, dgamma_m = 0.f, sc = 1.f, sh = 0.f;
  if (PHASE != 0) {
    mu = p.mean[c];
    rs = p.rstd[c];
    ga = p.gamma ? p.gamma[c] : 1.0f;
    be = p.beta ? p.beta[c] : 0.0f;
    // the forward's per-channel (scale, shift), bit for bit (bn_affine is what bn_final_kernel
    // stores): the PReLU side of an element is decided by the SAME fmaf(a, scale, shift) the
    // consuming contraction applied while staging it, so forward and backward agree on every
    // gate also where the normalised value is within roundoff of zero
    bn_affine(ga, be, mu, rs, sc, sh);
  }
  if (PHASE == 2) {
    const float* tot = p.ws + (size_t)p.nsplit * p.C * 4 + (size_t)c * 2;
    dbeta_m = tot[0] * p.inv_count;
    dgamma_m = tot[1] * p.inv_count;
  }
  float r[3] = {0.f, 0.f, 0.f};
  const bool has_skip = p.dskip != nullptr;
  // one element: returns the gradient written to da (phases 0 and 2), accumulates r
  auto elem = [&](float av, float dh, float ds) -> float {
    if (PHASE == 0) {
      float g = dh * (av > 0.f ? 1.0f : sl);
      r[0] += dh * (av > 0.f ? 0.0f : av);
      if (has_skip) {
        g = fmaf(al, ds, g);
        r[1] = fmaf(ds, av, r[1]);
      }
      r[2] += g;
      return g;
    }
    const float xh = (av - mu) * rs;
    const float v = fmaf(av, sc, sh);
    const float g = dh * (v > 0.f ? 1.0f : sl);
    if (PHASE == 1) {
      r[0] += dh * (v > 0.f ? 0.0f : v);
      r[1] += g;
      r[2] = fmaf(g, xh, r[2]);
      return 0.0f;
    }
    const float d = ga * rs * (g - dbeta_m - xh * dgamma_m);
    r[0] += d;
    return d;
  };
  // rows (b, c, :) of the slice: `tr` threads walk a row (float4 when 4 | L), PW_THREADS/tr
  // rows at a time; 32-bit index math only
  const size_t base = ((size_t)cr.b_beg * p.C + c) * p.L;
  const size_t sstride = (size_t)p.C * p.L;
  const bool vec = (p.L & 3) == 0 &&
                   (((uintptr_t)p.a | (uintptr_t)p.dh | (uintptr_t)p.dskip | (uintptr_t)p.da) & 15) == 0;
  const int Lv = vec ? p.L >> 2 : p.L;
  int tr = PW_THREADS;
  while (tr > Lv && tr > 1) tr >>= 1;
  const int rp = PW_THREADS / tr;
  const int t0 = threadIdx.x % tr, r0 = threadIdx.x / tr;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int b = r0; b < nb; b += rp) {
    const size_t ro = base + b * sstride;
    if (vec) {
      for (int t = t0; t < Lv; t += tr) {
        const size_t i = ro + 4 * (size_t)t;
        const f32x4 av = *reinterpret_cast<const f32x4*>(p.a + i);
        const f32x4 dh = p.dh ? *reinterpret_cast<const f32x4*>(p.dh + i) : z4;
        const f32x4 ds = has_skip ? *reinterpret_cast<const f32x4*>(p.dskip + i) : z4;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = elem(av[e], dh[e], ds[e]);
        if (PHASE != 1) *reinterpret_cast<f32x4*>(p.da + i) = o;
      }
    } else {
      for (int t = t0; t < Lv; t += tr) {
        const size_t i = ro + t;
        const float o = elem(p.a[i], p.dh ? p.dh[i] : 0.0f, has_skip ? p.dskip[i] : 0.0f);
        if (PHASE != 1) p.da[i] = o;
      }
    }
  }
  block_sum<3>(r, sm);
  if (threadIdx.x == 0) {
    float* w = p.ws + ((size_t)blockIdx.y * p.C + c) * 4;
    w[0] = r[0];
    w[1] = r[1];
    w[2] = r[2];
  }
}

// one thread per channel: sum partials, accumulate into the parameter gradients
template <int PHASE>
__global__ void act_bwd_final_kernel(const ActBwdArgs p, float* dslope, float* dalpha,
                                     float* dgamma, float* dbeta, float* dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int s = 0; s < p.nsplit; ++s) {
    const float* w = p.ws + ((size_t)s * p.C + c) * 4;
    s0 += w[0];
    s1 += w[1];
    s2 += w[2];
  }
  if (PHASE == 0) {
    if (dslope) dslope[c] += s0;
    if (dalpha) dalpha[c] += s1;
    if (dbias) dbias[c] += s2;
  } else if (PHASE == 1) {
    if (dslope) dslope[c] += s0;
    if (dbeta) dbeta[c] += s1;
    if (dgamma) dgamma[c] += s2;
    float* tot = p.ws + (size_t)p.nsplit * p.C * 4 + (size_t)c * 2;
    tot[0] = s1;
    tot[1] = s2;
  } else {
    if (dbias) dbias[c] += s0;
  }
}

extern "C" int segan_act_bwd(const float* a, const float* dh, const float* dskip,
                             const float* slope, const float* alpha, const float* bn_mean,
                             const float* bn_rstd, const float* bn_gamma, const float* bn_beta,
                             float* da, float* dslope, float* dalpha, float* dgamma, float* dbeta,
                             float* dbias, float* ws, int B, int C, int L, void* stream) {
  SEGAN_REQUIRE(a && da && ws, "act_bwd: NULL pointer");
  SEGAN_REQUIRE(dh || dskip, "act_bwd: no incoming gradient");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "act_bwd: bad sizes");
  SEGAN_REQUIRE((bn_mean == nullptr) == (bn_rstd == nullptr), "act_bwd: mean/rstd must pair");
  SEGAN_REQUIRE(!(dskip && !alpha), "act_bwd: dskip needs alpha");
  SEGAN_REQUIRE(!(bn_mean && dskip), "act_bwd: BN layers have no skip tap");
  hipStream_t st = (hipStream_t)stream;
  ActBwdArgs p;
  p.a = a; p.dh = dh; p.dskip = dskip; p.slope = slope; p.alpha = alpha;
  p.mean = bn_mean; p.rstd = bn_rstd; p.gamma = bn_gamma; p.beta = bn_beta;
  p.da = da; p.ws = ws; p.B = B; p.C = C; p.L = L;
  p.nsplit = pw_nsplit(B, C, L);
  p.inv_count = 1.0f / ((float)B * (float)L);
  const dim3 grid(C, p.nsplit), fgrid(ceil_div(C, 64));
  if (!bn_mean) {
    hipLaunchKernelGGL(act_bwd_kernel<0>, grid, dim3(PW_THREADS), 0, st, p);
    hipLaunchKernelGGL(act_bwd_final_kernel<0>, fgrid, dim3(64), 0, st, p, dslope, dalpha, dgamma,
                       dbeta, dbias);
  } else {
    hipLaunchKernelGGL(act_bwd_kernel<1>, grid, dim3(PW_THREADS), 0, st, p);
    hipLaunchKernelGGL(act_bwd_final_kernel<1>, fgrid, dim3(64), 0, st, p, dslope, dalpha, dgamma,
                       dbeta, dbias);
    hipLaunchKernelGGL(act_bwd_kernel<2>, grid, dim3(PW_THREADS), 0, st, p);
    hipLaunchKernelGGL(act_bwd_final_kernel<2>, fgrid, dim3(64), 0, st, p, dslope, dalpha, dgamma,
                       dbeta, dbias);
  }
  return segan_check_launch("act_bwd");
}

// The BatchNorm branch of segan_act_bwd in two calls, so that a data-parallel run can sum the
// per-channel totals over the ranks between them (synchronised BatchNorm): `reduce` leaves
// (sum g, sum g*xhat) per channel in totals[C][2]; `apply` consumes them with the GLOBAL
// element count per channel.  ws as for segan_act_bwd.
extern "C" int segan_act_bwd_bn_reduce(const float* a, const float* dh, const float* slope,
                                       const float* bn_mean, const float* bn_rstd,
                                       const float* bn_gamma, const float* bn_beta, float* dslope,
                                       float* dgamma, float* dbeta, float* totals, float* ws, int B,
                                       int C, int L, void* stream) {
  SEGAN_REQUIRE(a && dh && bn_mean && bn_rstd && totals && ws, "act_bwd_bn_reduce: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "act_bwd_bn_reduce: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  ActBwdArgs p;
  p.a = a; p.dh = dh; p.dskip = nullptr; p.slope = slope; p.alpha = nullptr;
  p.mean = bn_mean; p.rstd = bn_rstd; p.gamma = bn_gamma; p.beta = bn_beta;
  p.da = nullptr; p.ws = ws; p.B = B; p.C = C; p.L = L;
  p.nsplit = pw_nsplit(B, C, L);
  p.inv_count = 1.0f / ((float)B * (float)L);
  hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(C, p.nsplit), dim3(PW_THREADS), 0, st, p);
  hipLaunchKernelGGL(act_bwd_final_kernel<1>, dim3(ceil_div(C, 64)), dim3(64), 0, st, p, dslope,
                     (float*)nullptr, dgamma, dbeta, (float*)nullptr);
  if (hipMemcpyAsync(totals, ws + (size_t)p.nsplit * C * 4, (size_t)C * 2 * sizeof(float),
                     hipMemcpyDeviceToDevice, st) != hipSuccess) {
    segan_set_error("act_bwd_bn_reduce: copy of the totals failed");
    return SEGAN_ELAUNCH;
  }
  return segan_check_launch("act_bwd_bn_reduce");
}

extern "C" int segan_act_bwd_bn_apply(const float* a, const float* dh, const float* slope,
                                      const float* bn_mean, const float* bn_rstd,
                                      const float* bn_gamma, const float* bn_beta,
                                      const float* totals, float* da, float* dbias, float* ws, int B,
                                      int C, int L, double count_total, void* stream) {
  SEGAN_REQUIRE(a && dh && bn_mean && bn_rstd && totals && da && ws, "act_bwd_bn_apply: NULL pointer");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0 && count_total >= (double)B * L, "act_bwd_bn_apply: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  ActBwdArgs p;
  p.a = a; p.dh = dh; p.dskip = nullptr; p.slope = slope; p.alpha = nullptr;
  p.mean = bn_mean; p.rstd = bn_rstd; p.gamma = bn_gamma; p.beta = bn_beta;
  p.da = da; p.ws = ws; p.B = B; p.C = C; p.L = L;
  p.nsplit = pw_nsplit(B, C, L);
  p.inv_count = (float)(1.0 / count_total);
  if (hipMemcpyAsync(ws + (size_t)p.nsplit * C * 4, totals, (size_t)C * 2 * sizeof(float),
                     hipMemcpyDeviceToDevice, st) != hipSuccess) {
    segan_set_error("act_bwd_bn_apply: copy of the totals failed");
    return SEGAN_ELAUNCH;
  }
  hipLaunchKernelGGL(act_bwd_kernel<2>, dim3(C, p.nsplit), dim3(PW_THREADS), 0, st, p);
  hipLaunchKernelGGL(act_bwd_final_kernel<2>, dim3(ceil_div(C, 64)), dim3(64), 0, st, p,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, dbias);
  return segan_check_launch("act_bwd_bn_apply");
}

// ---------------------------------------------------------------------------------
// tanh backward (+ fused L1 term)
// ---------------------------------------------------------------------------------
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* dy, const float* clean,
                                float l1_scale, float* __restrict__ da, float* ws, int B, int C,
                                int L, int nsplit) {
  __shared__ float sm[16];
  const ChanRange cr = chan_range(B, nsplit);
  const int nb = cr.b_end - cr.b_beg;
  const long cnt = (long)(nb > 0 ? nb : 0) * L;
  float r[1] = {0.f};
  for (long e = threadIdx.x; e < cnt; e += PW_THREADS) {
    const int b = cr.b_beg + (int)(e / L);
    const int t = (int)(e % L);
    const size_t i = ((size_t)b * C + cr.c) * L + t;
    const float yv = y[i];
    float g = dy ? dy[i] : 0.0f;
    if (clean) {
      const float d = yv - clean[i];
      g += l1_scale * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f));
    }
    const float d = g * (1.0f - yv * yv);
    da[i] = d;
    r[0] += d;
  }
  block_sum<1>(r, sm);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.y * C + cr.c] = r[0];
}

__global__ void sum_splits_add_kernel(const float* ws, float* out, int C, int nsplit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += ws[(size_t)k * C + c];
  out[c] += s;
}

extern "C" int segan_tanh_bwd(const float* y, const float* dy, const float* clean, float l1_scale,
                              float* da, float* dbias, float* ws, int B, int C, int L,
                              void* stream) {
  SEGAN_REQUIRE(y && da && ws, "tanh_bwd: NULL pointer");
  SEGAN_REQUIRE(dy || clean, "tanh_bwd: no incoming gradient");
  SEGAN_REQUIRE(B > 0 && C > 0 && L > 0, "tanh_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int ns = pw_nsplit(B, C, L);
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(C, ns), dim3(PW_THREADS), 0, st, y, dy, clean, l1_scale,
                     da, ws, B, C, L, ns);
  if (dbias)
    hipLaunchKernelGGL(sum_splits_add_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, ws, dbias, C,
                       ns);
  return segan_check_launch("tanh_bwd");
}

// ---------------------------------------------------------------------------------
// dense head: bias + PReLU over [rows, cols]
// ---------------------------------------------------------------------------------
__global__ void bias_prelu_rows_kernel(const float* __restrict__ x, const float* bias,
                                       const float* slope, float* __restrict__ y, int rows,
                                       int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    float v = x[i] + (bias ? bias[c] : 0.0f);
    if (slope) v = v > 0.f ? v : v * slope[c];
    y[i] = v;
  }
}

extern "C" int segan_bias_prelu_rows(const float* x, const float* bias, const float* slope,
                                     float* y, int rows, int cols, void* stream) {
  SEGAN_REQUIRE(x && y && rows > 0 && cols > 0, "bias_prelu_rows: bad arguments");
  const size_t total = (size_t)rows * cols;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(bias_prelu_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x,
                     bias, slope, y, rows, cols);
  return segan_check_launch("bias_prelu_rows");
}

// one 256-thread workgroup per 64 columns: 4 row-groups walk the rows in parallel and are
// combined through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void bias_prelu_rows_bwd_kernel(
    const float* __restrict__ x, const float* bias, const float* slope,
    const float* __restrict__ dy, float* __restrict__ dx, float* dslope, float* dbias, int rows,
    int cols) {
  __shared__ float sm[2][4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s_sl = 0.f, s_b = 0.f;
  if (c < cols) {
    const float bs = bias ? bias[c] : 0.0f;
    const float sl = slope ? slope[c] : 1.0f;
    for (int r = rg; r < rows; r += 4) {
      const size_t i = (size_t)r * cols + c;
      const float v = x[i] + bs;
      const float g = dy[i];
      const float d = g * (v > 0.f ? 1.0f : sl);
      s_sl += g * (v > 0.f ? 0.0f : v);
      s_b += d;
      dx[i] = d;
    }
  }
  sm[0][rg][cl] = s_sl;
  sm[1][rg][cl] = s_b;
  __syncthreads();
  if (rg == 0 && c < cols) {
    const float a = sm[0][0][cl] + sm[0][1][cl] + sm[0][2][cl] + sm[0][3][cl];
    const float b = sm[1][0][cl] + sm[1][1][cl] + sm[1][2][cl] + sm[1][3][cl];
    if (dslope && slope) dslope[c] += a;
    if (dbias) dbias[c] += b;
  }
}

extern "C" int segan_bias_prelu_rows_bwd(const float* x, const float* bias, const float* slope,
                                         const float* dy, float* dx, float* dslope, float* dbias,
                                         int rows, int cols, void* stream) {
  SEGAN_REQUIRE(x && dy && dx && rows > 0 && cols > 0, "bias_prelu_rows_bwd: bad arguments");
  hipLaunchKernelGGL(bias_prelu_rows_bwd_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0,
                     (hipStream_t)stream, x, bias, slope, dy, dx, dslope, dbias, rows, cols);
  return segan_check_launch("bias_prelu_rows_bwd");
}

// ---------------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------------
__global__ void mse_const_kernel(const float* __restrict__ x, float target, float* loss,
                                 float* grad, const float* gout, float gscale, int n) {
  __shared__ float sm[16];
  float r[1] = {0.f};
  const float inv = 1.0f / (float)n;
  if (gout) gscale *= gout[0];
  for (int i = threadIdx.x; i < n; i += PW_THREADS) {
    const float d = x[i] - target;
    r[0] = fmaf(d, d, r[0]);
    if (grad) grad[i] = 2.0f * d * inv * gscale;
  }
  block_sum<1>(r, sm);
  if (threadIdx.x == 0 && loss) loss[0] = r[0] * inv;
}

extern "C" int segan_mse_const(const float* x, float target, float* loss, float* grad,
                               const float* gout, float gscale, int n, void* stream) {
  SEGAN_REQUIRE(x && n > 0 && (loss || grad), "mse_const: bad arguments");
  hipLaunchKernelGGL(mse_const_kernel, dim3(1), dim3(PW_THREADS), 0, (hipStream_t)stream, x, target,
                     loss, grad, gout, gscale, n);
  return segan_check_launch("mse_const");
}

// F.binary_cross_entropy_with_logits against a constant label (WSEGAN --vanilla_gan,
// model.py:582-583): loss = mean(max(x,0) - x*t + log1p(exp(-|x|))), d/dx = (sigmoid(x)-t)/n
__global__ void bce_const_kernel(const float* __restrict__ x, float target, float* loss,
                                 float* grad, const float* gout, float gscale, int n) {
  __shared__ float sm[16];
  float r[1] = {0.f};
  const float inv = 1.0f / (float)n;
  if (gout) gscale *= gout[0];
  for (int i = threadIdx.x; i < n; i += PW_THREADS) {
    const float v = x[i];
    r[0] += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    if (grad) grad[i] = (1.0f / (1.0f + expf(-v)) - target) * inv * gscale;
  }
  block_sum<1>(r, sm);
  if (threadIdx.x == 0 && loss) loss[0] = r[0] * inv;
}

extern "C" int segan_bce_logits_const(const float* x, float target, float* loss, float* grad,
                                      const float* gout, float gscale, int n, void* stream) {
  SEGAN_REQUIRE(x && n > 0 && (loss || grad), "bce_logits_const: bad arguments");
  hipLaunchKernelGGL(bce_const_kernel, dim3(1), dim3(PW_THREADS), 0, (hipStream_t)stream, x, target,
                     loss, grad, gout, gscale, n);
  return segan_check_launch("bce_logits_const");
}

// SQ = false: sum |x - y| (F.l1_loss); SQ = true: sum (x - y)^2 (F.mse_loss, --reg_loss mse_loss)
template <bool SQ>
__global__ void l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                  float* ws, size_t n) {
  __shared__ float sm[16];
  float r[1] = {0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - y[i];
    r[0] += SQ ? d * d : fabsf(d);
  }
  block_sum<1>(r, sm);
  if (threadIdx.x == 0) ws[blockIdx.x] = r[0];
}
__global__ void l1_final_kernel(const float* ws, float* loss, int nblocks, float inv) {
  __shared__ float sm[16];
  float r[1] = {0.f};
  for (int i = threadIdx.x; i < nblocks; i += PW_THREADS) r[0] += ws[i];
  block_sum<1>(r, sm);
  if (threadIdx.x == 0) loss[0] = r[0] * inv;
}

extern "C" int segan_l1_mean(const float* x, const float* y, float* loss, float* ws, int64_t n,
                             void* stream) {
  SEGAN_REQUIRE(x && y && loss && ws && n > 0, "l1_mean: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(l1_partial_kernel<false>, dim3(blocks), dim3(PW_THREADS), 0, st, x, y, ws, (size_t)n);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(PW_THREADS), 0, st, ws, loss, blocks,
                     1.0f / (float)n);
  return segan_check_launch("l1_mean");
}

extern "C" int segan_mse_mean(const float* x, const float* y, float* loss, float* ws, int64_t n,
                              void* stream) {
  SEGAN_REQUIRE(x && y && loss && ws && n > 0, "mse_mean: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(l1_partial_kernel<true>, dim3(blocks), dim3(PW_THREADS), 0, st, x, y, ws, (size_t)n);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(PW_THREADS), 0, st, ws, loss, blocks,
                     1.0f / (float)n);
  return segan_check_launch("mse_mean");
}

__global__ void mse_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                               const float* gout, float gscale, float* __restrict__ grad, size_t n) {
  const float g = 2.0f * gscale * (gout ? gout[0] : 1.0f) / (float)n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    grad[i] = g * (x[i] - y[i]);
}

extern "C" int segan_mse_bwd(const float* x, const float* y, const float* gout, float gscale,
                             float* grad, int64_t n, void* stream) {
  SEGAN_REQUIRE(x && y && grad && n > 0, "mse_bwd: bad arguments");
  int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, gout,
                     gscale, grad, (size_t)n);
  return segan_check_launch("mse_bwd");
}

__global__ void l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                              const float* gout, float gscale, float* __restrict__ grad, size_t n) {
  const float g = gscale * (gout ? gout[0] : 1.0f) / (float)n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - y[i];
    grad[i] = d > 0.f ? g : (d < 0.f ? -g : 0.0f);
  }
}

extern "C" int segan_l1_bwd(const float* x, const float* y, const float* gout, float gscale,
                            float* grad, int64_t n, void* stream) {
  SEGAN_REQUIRE(x && y && grad && n > 0, "l1_bwd: bad arguments");
  int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, gout,
                     gscale, grad, (size_t)n);
  return segan_check_launch("l1_bwd");
}

// ---------------------------------------------------------------------------------
// optimizers over a flat fp32 arena
// ---------------------------------------------------------------------------------
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                               float* __restrict__ sq, float lr, float alpha, float eps, size_t n) {
  const size_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* s4 = reinterpret_cast<float4*>(sq);
  const float oma = 1.0f - alpha;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 pv = p4[i], gv = g4[i], sv = s4[i];
#define RMS1(f)                                       \
  sv.f = alpha * sv.f + oma * gv.f * gv.f;            \
  pv.f = pv.f - lr * (gv.f / (sqrtf(sv.f) + eps));
    RMS1(x) RMS1(y) RMS1(z) RMS1(w)
#undef RMS1
    p4[i] = pv;
    s4[i] = sv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float gv = g[i];
    const float sv = alpha * sq[i] + oma * gv * gv;
    sq[i] = sv;
    p[i] = p[i] - lr * (gv / (sqrtf(sv) + eps));
  }
}

extern "C" int segan_rmsprop_step(float* p, const float* g, float* sq, float lr, float alpha,
                                  float eps, int64_t n, void* stream) {
  SEGAN_REQUIRE(p && g && sq && n > 0, "rmsprop_step: bad arguments");
  SEGAN_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)sq % 16 == 0),
                "rmsprop_step: arenas must be 16-byte aligned");
  int blocks = (int)((n / 4 + 255) / 256 > 2048 ? 2048 : (n / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, sq, lr,
                     alpha, eps, (size_t)n);
  return segan_check_launch("rmsprop_step");
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, float step_size,
                            float beta1, float beta2, float eps, float inv_sqrt_bc2, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float gv = g[i];
    // torch: exp_avg.lerp_(grad, 1-beta1); exp_avg_sq = beta2*v + (1-beta2)*g*g
    const float mv = m[i] + (gv - m[i]) * (1.0f - beta1);
    const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    p[i] = p[i] - step_size * (mv / denom);
  }
}

extern "C" int segan_adam_step(float* p, const float* g, float* m, float* v, float lr, float beta1,
                               float beta2, float eps, int step, int64_t n, void* stream) {
  SEGAN_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     step_size, beta1, beta2, eps, inv_sqrt_bc2, (size_t)n);
  return segan_check_launch("adam_step");
}

// ---------------------------------------------------------------------------------
// global pooling over time: the 'gmax' / 'gavg' discriminator heads
// (discriminator.py:128-137,183-190: AdaptiveMaxPool1d(1) / AdaptiveAvgPool1d(1))
// ---------------------------------------------------------------------------------
// one wavefront per row of [rows][L]; MODE 0: max and the FIRST position that attains it
// (where the gradient goes), MODE 1: mean
template <int MODE>
__global__ void pool_time_kernel(const float* __restrict__ x, float* __restrict__ y,
                                 int* __restrict__ idx, int rows, int L) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * L;
  if (MODE == 0) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = lane; t < L; t += 64) {
      const float v = xr[t];
      if (v > best || bi == 0x7fffffff) { best = v; bi = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      y[row] = best;
      idx[row] = bi;
    }
  } else {
    float s = 0.0f;
    for (int t = lane; t < L; t += 64) s += xr[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s / (float)L;
  }
}

template <int MODE>
__global__ void pool_time_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                     float* __restrict__ dx, int rows, int L) {
  const size_t n = (size_t)rows * L;
  const float invL = 1.0f / (float)L;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / L);
    const int t = (int)(i - (size_t)row * L);
    dx[i] = MODE == 0 ? (t == idx[row] ? dy[row] : 0.0f) : dy[row] * invL;
  }
}

extern "C" int segan_pool_time_fwd(const float* x, float* y, int* idx, int rows, int L, int mode,
                                   void* stream) {
  SEGAN_REQUIRE(x && y && rows > 0 && L > 0, "pool_time_fwd: bad arguments");
  SEGAN_REQUIRE(mode == 0 || mode == 1, "pool_time_fwd: mode must be 0 (max) or 1 (mean)");
  SEGAN_REQUIRE(mode == 1 || idx, "pool_time_fwd: max pooling needs idx");
  const dim3 grid(ceil_div(rows, 4));
  if (mode == 0)
    hipLaunchKernelGGL(pool_time_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, y, idx,
                       rows, L);
  else
    hipLaunchKernelGGL(pool_time_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, y, idx,
                       rows, L);
  return segan_check_launch("pool_time_fwd");
}

extern "C" int segan_pool_time_bwd(const float* dy, const int* idx, float* dx, int rows, int L,
                                   int mode, void* stream) {
  SEGAN_REQUIRE(dy && dx && rows > 0 && L > 0, "pool_time_bwd: bad arguments");
  SEGAN_REQUIRE(mode == 0 || mode == 1, "pool_time_bwd: mode must be 0 (max) or 1 (mean)");
  SEGAN_REQUIRE(mode == 1 || idx, "pool_time_bwd: max pooling needs idx");
  const size_t n = (size_t)rows * L;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (mode == 0)
    hipLaunchKernelGGL(pool_time_bwd_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       dy, idx, dx, rows, L);
  else
    hipLaunchKernelGGL(pool_time_bwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       dy, idx, dx, rows, L);
  return segan_check_launch("pool_time_bwd");
}

__global__ void fill_kernel(float* p, float value, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = value;
}
__global__ void scale_kernel(float* p, float s, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] *= s;
}
extern "C" int segan_fill(float* p, float value, int64_t n, void* stream) {
  SEGAN_REQUIRE(p && n > 0, "fill: bad arguments");
  int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, value,
                     (size_t)n);
  return segan_check_launch("fill");
}
extern "C" int segan_scale(float* p, float s, int64_t n, void* stream) {
  SEGAN_REQUIRE(p && n > 0, "scale: bad arguments");
  int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, s,
                     (size_t)n);
  return segan_check_launch("scale");
}
