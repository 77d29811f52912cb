// segan_snorm.hip — spectral normalisation of a weight tensor, the 'snorm' norm type of the
// reference (modules.py:12-14, discriminator.py:118-121: torch.nn.utils.spectral_norm with its
// defaults n_power_iterations = 1, eps = 1e-12):
//
//   W_mat = weight viewed as [rows, cols] (rows = dim 0 for Conv1d / Linear / PReLU,
//           dim 1 for ConvTranspose1d)
//   training:  v <- normalize(W_mat^T u);  u <- normalize(W_mat v)      (in place, no grad)
//   sigma = u . (W_mat v);   W_sn = W / sigma
//   backward (u, v constants):  dW += dW_sn / sigma - (<dW_sn, W> / sigma^2) * u v^T
//
// The weight is [A][Bd][K] contiguous (Linear: K = 1; PReLU: Bd = K = 1).  All of it is
// HBM-bound streaming over the weight (<= 130 MB); a forward reads it three times.
#include "segan_common.h"

struct SnView {
  int A, Bd, K, dim;
  int rows, cols;
};

// element (r, c) of the matrix view -> linear index into the weight
__device__ __forceinline__ size_t sn_index(const SnView& s, int r, int c) {
  if (s.dim == 0) return (size_t)r * s.cols + c;
  const int a = c / s.K, k = c - a * s.K;
  return ((size_t)a * s.Bd + r) * s.K + k;
}

// t[c] = sum_r W(r, c) * u[r]; one thread per column, rows split over blockIdx.y (atomics)
__global__ void sn_matvec_t_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                   float* __restrict__ t, SnView s, int rows_per) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= s.cols) return;
  const int r0 = blockIdx.y * rows_per, r1 = min(s.rows, r0 + rows_per);
  float acc = 0.0f;
  for (int r = r0; r < r1; ++r) acc = fmaf(w[sn_index(s, r, c)], u[r], acc);
  atomicAdd(t + c, acc);
}

// sv[r] = sum_c W(r, c) * v[c]; one workgroup per row
__global__ void sn_matvec_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                 float* __restrict__ sv, SnView s) {
  __shared__ float sm[4];
  const int r = blockIdx.x;
  float acc = 0.0f;
  for (int c = threadIdx.x; c < s.cols; c += blockDim.x) acc = fmaf(w[sn_index(s, r, c)], v[c], acc);
  acc = warp_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sv[r] = sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ float sn_block_sum(float v, float* sm) {
  v = warp_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float tot = 0.0f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += sm[i];
  return tot;
}

// dst = src / max(||src||, eps); optionally sigma[0] = dst . src  (single workgroup)
__global__ void sn_normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, int n,
                                    float eps, float* sigma) {
  __shared__ float sm[16];
  float q = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) q = fmaf(src[i], src[i], q);
  const float nrm = sqrtf(sn_block_sum(q, sm));
  const float inv = 1.0f / fmaxf(nrm, eps);
  float d = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = src[i];
    const float y = x * inv;
    dst[i] = y;
    d = fmaf(y, x, d);
  }
  if (sigma) {
    const float tot = sn_block_sum(d, sm);
    if (threadIdx.x == 0) sigma[0] = tot;
  }
}

// sigma[0] = a . b  (single workgroup)
__global__ void sn_dot_small_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                    float* sigma) {
  __shared__ float sm[16];
  float d = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) d = fmaf(a[i], b[i], d);
  const float tot = sn_block_sum(d, sm);
  if (threadIdx.x == 0) sigma[0] = tot;
}

__global__ void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                float* __restrict__ out, size_t n) {
  const float inv = 1.0f / sigma[0];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = w[i] * inv;
}

// partial[blockIdx.x] = sum over the block's grid-stride slice of a[i]*b[i]
__global__ void sn_dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                      size_t n, float* __restrict__ partial) {
  __shared__ float sm[16];
  float d = 0.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d = fmaf(a[i], b[i], d);
  const float tot = sn_block_sum(d, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// dw[i] += dw_sn[i]/sigma - (dot/sigma^2) * u[r] * v[c]
__global__ void sn_bwd_kernel(const float* __restrict__ dw_sn, const float* __restrict__ u,
                              const float* __restrict__ v, const float* __restrict__ sigma,
                              const float* __restrict__ partial, int npartial, float* __restrict__ dw,
                              SnView s, size_t n) {
  __shared__ float s_dot;
  if (threadIdx.x == 0) {
    float d = 0.0f;
    for (int i = 0; i < npartial; ++i) d += partial[i];
    s_dot = d;
  }
  __syncthreads();
  const float inv = 1.0f / sigma[0];
  const float coef = s_dot * inv * inv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (s.dim == 0) {
      r = (int)(i / s.cols);
      c = (int)(i - (size_t)r * s.cols);
    } else {
      const int k = (int)(i % s.K);
      const size_t ab = i / s.K;
      r = (int)(ab % s.Bd);
      c = (int)(ab / s.Bd) * s.K + k;
    }
    dw[i] += dw_sn[i] * inv - coef * u[r] * v[c];
  }
}

static int sn_view(SnView* s, int A, int Bd, int K, int dim, const char* what) {
  if (A <= 0 || Bd <= 0 || K <= 0 || (dim != 0 && dim != 1)) {
    segan_set_error("%s: bad weight view [%d][%d][%d] dim %d", what, A, Bd, K, dim);
    return SEGAN_EINVAL;
  }
  s->A = A; s->Bd = Bd; s->K = K; s->dim = dim;
  s->rows = dim == 0 ? A : Bd;
  s->cols = dim == 0 ? Bd * K : A * K;
  return SEGAN_OK;
}

#define SN_PARTIALS 1024

extern "C" size_t segan_snorm_ws_floats(int A, int Bd, int K, int dim) {
  if (A <= 0 || Bd <= 0 || K <= 0) return 0;
  const size_t rows = dim == 0 ? A : Bd, cols = dim == 0 ? (size_t)Bd * K : (size_t)A * K;
  return rows + cols + SN_PARTIALS;
}

extern "C" int segan_snorm_fwd(const float* w, float* u, float* v, float* w_sn, float* sigma,
                               float* ws, int A, int Bd, int K, int dim, int power_iteration,
                               float eps, void* stream) {
  SEGAN_REQUIRE(w && u && v && w_sn && sigma && ws, "snorm_fwd: NULL pointer");
  SnView s;
  if (int e = sn_view(&s, A, Bd, K, dim, "snorm_fwd")) return e;
  hipStream_t st = (hipStream_t)stream;
  float* t = ws;            // [cols]
  float* sv = ws + s.cols;  // [rows]
  if (power_iteration) {
    if (hipMemsetAsync(t, 0, (size_t)s.cols * sizeof(float), st) != hipSuccess) {
      segan_set_error("snorm_fwd: memset failed");
      return SEGAN_ELAUNCH;
    }
    int ysplit = ceil_div(1024 * 256, s.cols);           // enough workgroups for thin matrices
    if (ysplit > ceil_div(s.rows, 16)) ysplit = ceil_div(s.rows, 16);
    if (ysplit < 1) ysplit = 1;
    const int rows_per = ceil_div(s.rows, ysplit);
    hipLaunchKernelGGL(sn_matvec_t_kernel, dim3(ceil_div(s.cols, 256), ceil_div(s.rows, rows_per)),
                       dim3(256), 0, st, w, u, t, s, rows_per);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, t, v, s.cols, eps,
                       (float*)nullptr);
    hipLaunchKernelGGL(sn_matvec_kernel, dim3(s.rows), dim3(256), 0, st, w, v, sv, s);
    // u = normalize(W v); sigma = u . (W v)
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, sv, u, s.rows, eps, sigma);
  } else {
    hipLaunchKernelGGL(sn_matvec_kernel, dim3(s.rows), dim3(256), 0, st, w, v, sv, s);
    hipLaunchKernelGGL(sn_dot_small_kernel, dim3(1), dim3(1024), 0, st, u, sv, s.rows, sigma);
  }
  const size_t n = (size_t)A * Bd * K;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(sn_scale_kernel, dim3(blocks), dim3(256), 0, st, w, sigma, w_sn, n);
  return segan_check_launch("snorm_fwd");
}

extern "C" int segan_snorm_bwd(const float* dw_sn, const float* w, const float* u, const float* v,
                               const float* sigma, float* dw, float* ws, int A, int Bd, int K,
                               int dim, void* stream) {
  SEGAN_REQUIRE(dw_sn && w && u && v && sigma && dw && ws, "snorm_bwd: NULL pointer");
  SnView s;
  if (int e = sn_view(&s, A, Bd, K, dim, "snorm_bwd")) return e;
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)A * Bd * K;
  float* partial = ws + s.rows + s.cols;
  int nb = (int)((n + 255) / 256 > SN_PARTIALS ? SN_PARTIALS : (n + 255) / 256);
  hipLaunchKernelGGL(sn_dot_partial_kernel, dim3(nb), dim3(256), 0, st, dw_sn, w, n, partial);
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(blocks), dim3(256), 0, st, dw_sn, u, v, sigma, partial, nb,
                     dw, s, n);
  return segan_check_launch("snorm_bwd");
}
