// segan_pack.hip — polyphase packing of a weight tensor [m][n][K] into the zero-padded
// F and T operand layouts of the fp32 contraction kernels (layouts: DESIGN.md section 4,
// segan_pytorch_amd/layout.py).
#include "segan_conv_shared.h"

// ====================================================================================
// weight packing
// ====================================================================================
// Both packings are [*, K] -> [K', *] transposes of a 64 x 32 tile through LDS so that the
// global reads (31 contiguous taps per (m,n)) and the writes (64 contiguous m / n) are both
// coalesced.  grid.x = tiles of 64 along the transposed axis, grid.y = the other axis.
__global__ __launch_bounds__(256) void pack_f_kernel(const float* __restrict__ w,
                                                     float* __restrict__ wf, int M, int N, int K,
                                                     int S, int U, int pitch, int rows, int pair) {
  __shared__ float t[64][33];
  const int n = blockIdx.y;                 // may run past N into the zero padding rows
  const int m0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  // all 8 loads of a thread in flight before the first LDS store (left as a rolled loop the
  // compiler issues load - wait - store eight times in a row: latency-bound, 2.8 TB/s)
  {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      const int ml = e >> 5, k = e & 31;
      const int m = m0 + ml;
      v[i] = (m < M && n < N && k < K) ? w[((size_t)m * N + n) * K + k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      t[e >> 5][e & 31] = v[i];
    }
  }
  // f_pair (K = 31): row 31 = (r, u) = (S-1, U-1), the padding tap k = 31, of an ODD channel
  // holds row 30 = (S-1, U-2), i.e. tap k = 31 - S, of its even partner n - 1 (segan_conv_shared.h)
  __syncthreads();
  if (pair && (n & 1) && n < N && tid < 64) {
    const int m = m0 + tid;
    t[tid][31] = m < M ? w[((size_t)m * N + n - 1) * K + (31 - S)] : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < 32 * 64; e += 256) {
    const int kk = e >> 6, ml = e & 63;     // kk = r*U + u  ->  tap k = S*u + r
    const int r = kk / U, u = kk - r * U;
    const int row = (n * S + r) * U + u;
    if (row < rows && m0 + ml < pitch) wf[(size_t)row * pitch + m0 + ml] = t[ml][S * u + r];
  }
}

__global__ __launch_bounds__(256) void pack_t_kernel(const float* __restrict__ w,
                                                     float* __restrict__ wt, int M, int N, int K,
                                                     int S, int U, int NP, int pad, int pitch,
                                                     int rows) {
  __shared__ float t[64][33];
  const int m = blockIdx.y;                 // may run past M into the zero padding rows
  const int n0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      const int nl = e >> 5, k = e & 31;
      const int n = n0 + nl;
      v[i] = (m < M && n < N && k < K) ? w[((size_t)m * N + n) * K + k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      t[e >> 5][e & 31] = v[i];
    }
  }
  __syncthreads();
  for (int e = tid; e < 32 * 64; e += 256) {
    const int kk = e >> 6, nl = e & 63;     // kk = u'*S + r
    const int up = kk / S, r = kk - up * S;
    const int rho = (r + pad) % S;
    const int k = S * (U - 1 - up) + rho;   // < 32 always; taps >= K hold zeros in t
    const int row = m * U + up;
    const int n = n0 + nl;
    if (row < rows && n < NP) wt[(size_t)row * pitch + r * NP + n] = t[nl][k];
  }
}

// ====================================================================================
// C ABI
// ====================================================================================
extern "C" size_t segan_packed_f_bytes(int M, int N, int S) {
  if (!stride_ok(S) || M <= 0 || N <= 0) return 0;
  return (size_t)f_rows(N) * f_pitch(M) * sizeof(float);
}
extern "C" size_t segan_packed_t_bytes(int M, int N, int S) {
  if (!stride_ok(S) || M <= 0 || N <= 0) return 0;
  return (size_t)t_rows(M, S) * t_pitch(N, S) * sizeof(float);
}

extern "C" int segan_pack_weights(const float* w, float* wf, float* wt, int M, int N, int K, int S,
                                  int pad_t, void* stream) {
  SEGAN_REQUIRE(w != nullptr, "pack_weights: w is NULL");
  SEGAN_REQUIRE(stride_ok(S), "pack_weights: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "pack_weights: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(M > 0 && N > 0, "pack_weights: bad channel counts %d,%d", M, N);
  SEGAN_REQUIRE(pad_t >= 0, "pack_weights: negative padding");
  hipStream_t st = (hipStream_t)stream;
  const int U = 32 / S;
  if (wf) {
    const int pitch = f_pitch(M), rows = f_rows(N);
    // rows = round_up(N*32, 64): cover the padding rows with one extra n when N is odd
    hipLaunchKernelGGL(pack_f_kernel, dim3(pitch / 64, ceil_div(rows, 32)), dim3(256), 0, st, w,
                       wf, M, N, K, S, U, pitch, rows, f_pair(N, K));
  }
  if (wt) {
    const int NP = t_np(N, S);
    const int pitch = t_pitch(N, S), rows = t_rows(M, S);
    hipLaunchKernelGGL(pack_t_kernel, dim3(ceil_div(NP, 64), ceil_div(rows, U)), dim3(256), 0, st,
                       w, wt, M, N, K, S, U, NP, pad_t, pitch, rows);
  }
  return segan_check_launch("pack_weights");
}
