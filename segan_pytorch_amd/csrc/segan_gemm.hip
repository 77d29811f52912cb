// segan_gemm.hip — small exact-fp32 MFMA GEMM for the discriminator's dense head
// (reference segan/models/discriminator.py:111-117: Linear 16384-256-128-1, forward
// and both gradients).  These are 0.2 % of the step's FLOPs; the kernel favours
// generality (arbitrary element strides, so no operand is ever transposed in HBM)
// over peak rate: 64x64 tiles, 4 waves of one 32x32 MFMA block each, split-K over
// blockIdx.z with fp32 atomics.
#include "segan_common.h"

#define GT 64
#define GK 32

__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, long sam, long sak,
                                                   const float* __restrict__ B, long sbk, long sbn,
                                                   float* __restrict__ C, long ldc, int M, int N,
                                                   int K, int kper) {
  __shared__ float Al[GT][GK + 1];
  __shared__ float Bl[GK][GT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    // A tile: 64 x 32 ; B tile: 32 x 64 (2048 elements each, 8 per thread)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      {
        // choose the fast-varying index to follow the smaller stride
        int r, kk;
        if (sak <= sam) { kk = e % GK; r = e / GK; } else { r = e % GT; kk = e / GT; }
        const int gm = m0 + r, gk = k0 + kk;
        Al[r][kk] = (gm < M && gk < kend) ? A[(long)gm * sam + (long)gk * sak] : 0.f;
      }
      {
        int c, kk;
        if (sbn <= sbk) { c = e % GT; kk = e / GT; } else { kk = e % GK; c = e / GK; }
        const int gn = n0 + c, gk = k0 + kk;
        Bl[kk][c] = (gn < N && gk < kend) ? B[(long)gk * sbk + (long)gn * sbn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      const float av = Al[wm * 32 + l31][2 * s + h];
      const float bv = Bl[2 * s + h][wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const bool atomic = gridDim.z > 1;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    const int n = n0 + wn * 32 + l31;
    if (m < M && n < N) {
      float* c = C + (long)m * ldc + n;
      if (atomic) atomicAdd(c, acc[e]);
      else *c += acc[e];
    }
  }
}

extern "C" int segan_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
                          int64_t sbn, float* C, int64_t ldc, int M, int N, int K, int beta0,
                          void* stream) {
  SEGAN_REQUIRE(A && B && C, "gemm: NULL pointer");
  SEGAN_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "gemm: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (beta0) {
    if (ldc == N) {
      if (hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), st) != hipSuccess) {
        segan_set_error("gemm: memset failed");
        return SEGAN_ELAUNCH;
      }
    } else {
      if (hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st) !=
          hipSuccess) {
        segan_set_error("gemm: memset2d failed");
        return SEGAN_ELAUNCH;
      }
    }
  }
  const int tm = ceil_div(M, GT), tn = ceil_div(N, GT);
  int nsplit = ceil_div(512, tm * tn);
  const int kchunks = ceil_div(K, GK);
  if (nsplit > kchunks) nsplit = kchunks;
  if (nsplit < 1) nsplit = 1;
  int kper = ceil_div(kchunks, nsplit) * GK;
  nsplit = ceil_div(K, kper);
  hipLaunchKernelGGL(gemm_kernel, dim3(tn, tm, nsplit), dim3(256), 0, st, A, (long)sam, (long)sak,
                     B, (long)sbk, (long)sbn, C, (long)ldc, M, N, K, kper);
  return segan_check_launch("gemm");
}
