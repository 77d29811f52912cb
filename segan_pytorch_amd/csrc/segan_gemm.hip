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


// ====================================================================================
// 128 x 128 x 16 tile kernel for operands with a unit-stride dimension (every GEMM of the
// dense head and of the STFT power loss has one): float4 global loads along that dimension,
// k-major LDS tiles so the MFMA fragments are conflict-free ds_read_b32, register prefetch of
// the next chunk, two LDS buffers, one barrier per chunk.  4 waves as 2 x 2, each 64 x 64.
// AK / BK: that operand is contiguous along k (else along m resp. n).
// ====================================================================================
#define G2T 128
#define G2K 16
#define G2P (G2T + 4)

template <bool AK, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const float* __restrict__ A, long lda,
                                                         const float* __restrict__ B, long ldb,
                                                         float* __restrict__ C, long ldc, int M,
                                                         int N, int K, int kper,
                                                         float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float Al[2][G2K][G2P];
  __shared__ __attribute__((aligned(16))) float Bl[2][G2K][G2P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * G2T, n0 = blockIdx.x * G2T;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging: 2048 elements per operand and chunk = two float4 per thread.
  //   k-contiguous operand: element (row r, k) at P[r*ld + k]; unit e -> (r = e / 4, k4 = e % 4)
  //   row-contiguous operand: element (row r, k) at P[k*ld + r]; unit e -> (k = e / 32, r4 = e % 32)
  f32x4 ra[2], rb[2];
  auto load = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (AK) {
        const int r = m0 + e / 4, k = k0 + 4 * (e % 4);
        if (r < M && k < kend) v = *reinterpret_cast<const f32x4*>(A + (long)r * lda + k);
        if (k + 3 >= kend) {   // ragged end of the k range (kend % 4 != 0 never happens: host)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (k + j < kend) ? v[j] : 0.f;
        }
      } else {
        const int k = k0 + e / 32, r = m0 + 4 * (e % 32);
        if (k < kend && r < M) v = *reinterpret_cast<const f32x4*>(A + (long)k * lda + r);
      }
      ra[i] = v;
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      if (BKC) {
        const int c = n0 + e / 4, k = k0 + 4 * (e % 4);
        if (c < N && k < kend) w = *reinterpret_cast<const f32x4*>(B + (long)c * ldb + k);
      } else {
        const int k = k0 + e / 32, c = n0 + 4 * (e % 32);
        if (k < kend && c < N) w = *reinterpret_cast<const f32x4*>(B + (long)k * ldb + c);
      }
      rb[i] = w;
    }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i;
      if (AK) {
        const int r = e / 4, k = 4 * (e % 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) Al[buf][k + j][r] = ra[i][j];
      } else {
        *reinterpret_cast<f32x4*>(&Al[buf][e / 32][4 * (e % 32)]) = ra[i];
      }
      if (BKC) {
        const int c = e / 4, k = 4 * (e % 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) Bl[buf][k + j][c] = rb[i][j];
      } else {
        *reinterpret_cast<f32x4*>(&Bl[buf][e / 32][4 * (e % 32)]) = rb[i];
      }
    }
  };
  if (kbeg >= kend) return;
  load(kbeg);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += G2K) {
    const bool more = k0 + G2K < kend;
    if (more) load(k0 + G2K);
#pragma unroll
    for (int s = 0; s < G2K / 2; ++s) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = Al[buf][2 * s + h][wm * 64 + 32 * i + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Bl[buf][2 * s + h][wn * 64 + 32 * j + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  const bool atomic = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + 32 * j + l31;
      const int mb = m0 + wm * 64 + 32 * i + 4 * h;
      if (!slabs && !atomic) {
        // unsplit: C += acc.  All 16 loads of the block before its first store: a load issued after a
        // store waits for that store (one in-order counter) — element by element this was 64 dependent
        // read-modify-write round trips per thread
        float old[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          old[e] = (m < M && n < N) ? C[(long)m * ldc + n] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          if (m < M && n < N) C[(long)m * ldc + n] = old[e] + acc[i][j][e];
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m < M && n < N) {
          if (slabs) {      // deterministic split-K: this split's partial, added in split order later
            slabs[((size_t)blockIdx.z * M + m) * N + n] = acc[i][j][e];
            continue;
          }
          atomicAdd(C + (long)m * ldc + n, acc[i][j][e]);
        }
      }
    }
}

// C (+)= sum over the contraction splits, in split order (bit-reproducible)
__global__ void gemm_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, long ldc,
                                   int M, int N, int nsplit, int beta0) {
  const size_t total = (size_t)M * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int z = 0; z < nsplit; ++z) s += slabs[(size_t)z * total + i];
    float* c = C + (i / N) * ldc + i % N;
    *c = beta0 ? s : *c + s;
  }
}

extern "C" size_t segan_gemm_scratch_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int tm = ceil_div(M, G2T), tn = ceil_div(N, G2T);
  int nsplit = ceil_div(512, tm * tn);
  const int kchunks = ceil_div(K, G2K);
  if (nsplit > kchunks / 4) nsplit = kchunks / 4;
  if (nsplit < 1) nsplit = 1;
  return (size_t)nsplit * M * N * sizeof(float);
}

extern "C" int segan_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
                          int64_t sbn, float* C, int64_t ldc, int M, int N, int K, int beta0,
                          int flags, void* scratch, size_t scratch_bytes, void* stream) {
  SEGAN_REQUIRE(A && B && C, "gemm: NULL pointer");
  SEGAN_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "gemm: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  // split-K partials are combined with fp32 atomics; with SEGAN_GEMM_DETERMINISTIC they are written
  // as slabs into the caller's scratch and added in split order (bit-reproducible) — without
  // scratch the whole contraction stays in one workgroup per tile
  const bool det = (flags & SEGAN_GEMM_DETERMINISTIC) != 0;
  const bool det_slabs = det && scratch != nullptr &&
                         scratch_bytes >= segan_gemm_scratch_bytes(M, N, K) && ((uintptr_t)scratch & 15) == 0;
  const bool nosplit = det && !det_slabs;
  if (beta0 && !det_slabs) {
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
  // fast path: both operands have a unit stride, 16-byte aligned rows, K a multiple of 4
  {
    const bool ak = sak == 1, am = sam == 1, bk = sbk == 1, bn = sbn == 1;
    const long lda = ak ? sam : sak, ldb = bk ? sbn : sbk;
    if ((ak || (am && M % 4 == 0)) && (bk || (bn && N % 4 == 0)) && K % 4 == 0 &&
        lda % 4 == 0 && ldb % 4 == 0 &&
        ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && (long)M * N >= 64 * 64) {
      const int tm = ceil_div(M, G2T), tn = ceil_div(N, G2T);
      int nsplit = ceil_div(512, tm * tn);
      const int kchunks = ceil_div(K, G2K);
      if (nsplit > kchunks / 4) nsplit = kchunks / 4;
      if (nsplit < 1 || nosplit) nsplit = 1;
      const int kper = ceil_div(kchunks, nsplit) * G2K;
      nsplit = ceil_div(K, kper);
      const dim3 grid(tn, tm, nsplit);
      float* slabs = (det_slabs && nsplit > 1) ? (float*)scratch : nullptr;
      if (det_slabs && nsplit == 1 && beta0) {      // unsplit: the kernel adds into a zeroed C
        if (hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st) != hipSuccess) {
          segan_set_error("gemm: memset2d failed");
          return SEGAN_ELAUNCH;
        }
      }
#define G2(AKF, BKF) hipLaunchKernelGGL((gemm128_kernel<AKF, BKF>), grid, dim3(256), 0, st, A, lda, B, ldb, C, (long)ldc, M, N, K, kper, slabs)
      // a k-contiguous view is preferred when both strides are 1 (degenerate 1-wide operands)
      if (ak && bk) G2(true, true); else if (ak) G2(true, false); else if (bk) G2(false, true); else G2(false, false);
#undef G2
      if (int e = segan_check_launch("gemm128")) return e;
      if (slabs) {
        const size_t total = (size_t)M * N;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)),
                           dim3(256), 0, st, slabs, C, (long)ldc, M, N, nsplit, beta0);
        return segan_check_launch("gemm_reduce");
      }
      return SEGAN_OK;
    }
  }
  if (beta0 && det_slabs) {     // (the generic kernel has no slab form: unsplit into a zeroed C)
    if (hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st) != hipSuccess) {
      segan_set_error("gemm: memset2d failed");
      return SEGAN_ELAUNCH;
    }
  }
  const int tm = ceil_div(M, GT), tn = ceil_div(N, GT);
  int nsplit = ceil_div(512, tm * tn);
  const int kchunks = ceil_div(K, GK);
  if (nsplit > kchunks) nsplit = kchunks;
  if (nsplit < 1 || det) nsplit = 1;
  int kper = ceil_div(kchunks, nsplit) * GK;
  nsplit = ceil_div(K, kper);
  hipLaunchKernelGGL(gemm_kernel, dim3(tn, tm, nsplit), dim3(256), 0, st, A, (long)sam, (long)sak,
                     B, (long)sbk, (long)sbn, C, (long)ldc, M, N, K, kper);
  return segan_check_launch("gemm");
}
