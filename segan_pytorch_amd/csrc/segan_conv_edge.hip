// segan_conv_edge.hip — direct VALU kernels for the HBM-bound edge layers (1-2 channels on
// one side), where an MFMA tile would be mostly padding: the T form for 1-2 output channels
// (last deconv of G; data gradient of the first conv) and the F form for 1-2 input channels
// (first conv of G and of D).  See segan_conv.hip for the two forms.
#include "segan_conv_shared.h"

// ====================================================================================
// T form for 1-2 output channels (the HBM-bound edge layers: the generator's last deconv
// Cout=1, and the data gradient of the first conv whose input has 1-2 channels).  With so
// few output channels an MFMA tile would be >90 % padding, so this is a direct VALU kernel:
// one thread per low-rate position q computes all S phases x N channels, the input window
// comes from an LDS tile (with the segan_src transform applied while staging) and the taps
// are wave-uniform scalar loads.
// ====================================================================================
// KT: kernel width known at compile time (31, the SEGAN width: the taps then sit at constant
// offsets and the scalar loads merge into s_load_dwordx8/x16) or 0 = runtime K.
template <int S, int N, int PM, int KT>
__global__ __launch_bounds__(256) void tsmall_kernel(const CorrArgs a, const float* __restrict__ w,
                                                     int Krt, int M) {
  const int K = KT ? KT : Krt;
  constexpr int U = 32 / S;
  constexpr int MC = 16;                 // input channels per LDS chunk
  constexpr int TW = 256 + U;            // window: 256 positions + (U-1) taps + 1 phase shift
  __shared__ float xs[MC][TW + 1];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 256;
  const int q = q0 + tid;
  // window coordinate j <-> input time t = q0 + win_start + j   (win_start = cmin - (U-1))
  float acc[S][N];
#pragma unroll
  for (int r = 0; r < S; ++r)
#pragma unroll
    for (int n = 0; n < N; ++n) acc[r][n] = 0.0f;

  // staging: thread owns window positions tid and 256 + tid (the latter only for tid < U);
  // addresses are clamped so the loads are unconditional
  const int t0 = q0 + a.win_start + tid, t1 = t0 + 256;
  const bool ok0 = t0 >= 0 && t0 < a.Lin, ok1 = tid < U && t1 >= 0 && t1 < a.Lin;
  const int o0 = ok0 ? t0 : 0, o1 = ok1 ? t1 : 0;
  const int bo0 = b * a.in.C0 * a.Lin, bo1 = b * a.in.C1 * a.Lin;
  // the next chunk's 16 x 2 loads are issued before the FMAs of the current chunk (round 6, as tsmall4)
  float pv0[MC], pv1[MC];
  auto load_rows = [&](int mc0) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      const int m = mc0 + mc < M ? mc0 + mc : 0;
      const bool seg1 = m >= a.in.C0;
      const float* rowp = seg1 ? a.in.p1 + (size_t)(m - a.in.C0) * a.Lin + bo1
                               : a.in.p0 + (size_t)m * a.Lin + bo0;
      pv0[mc] = rowp[o0];
      pv1[mc] = rowp[o1];
    }
  };
  load_rows(0);
  for (int mc0 = 0; mc0 < M; mc0 += MC) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      const bool mok = mc0 + mc < M;
      const ChanXf xf = segan_chan_xf(a.in, mok ? mc0 + mc : 0);
      xs[mc][tid] = (mok && ok0) ? segan_apply_xf(xf, pv0[mc]) : 0.0f;
      if (tid < U) xs[mc][256 + tid] = (mok && ok1) ? segan_apply_xf(xf, pv1[mc]) : 0.0f;
    }
    __syncthreads();
    if (mc0 + MC < M) load_rows(mc0 + MC);
    const int mcn = min(MC, M - mc0);
    for (int mc = 0; mc < mcn; ++mc) {
      float xv[U + 1];
#pragma unroll
      for (int j = 0; j <= U; ++j) xv[j] = xs[mc][tid + j];
      // taps are wave-uniform: scalar loads straight into SGPR operands of the FMAs; tap
      // indices are clamped and the value selected to zero for k >= K (no branches)
      const float* wm = w + (size_t)(mc0 + mc) * N * K;
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const int rho = (r + PM) % S;     // tap phase of output phase r
        const int cs = (r + PM) / S;      // 0/1: extra input shift of this phase
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = S * u + rho;
          if (KT) {
            if (k < KT) {
#pragma unroll
              for (int n = 0; n < N; ++n)
                acc[r][n] = fmaf(wm[n * KT + k], xv[cs + (U - 1) - u], acc[r][n]);
            }
          } else {
            const int kc = k < K ? k : K - 1;
#pragma unroll
            for (int n = 0; n < N; ++n) {
              float wv = wm[n * K + kc];
              wv = k < K ? wv : 0.0f;
              acc[r][n] = fmaf(wv, xv[cs + (U - 1) - u], acc[r][n]);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (q >= a.Tcols) return;
#pragma unroll
  for (int r = 0; r < S; ++r)
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float v = acc[r][n] + (a.bias ? a.bias[n] : 0.0f);
      if (a.act == SEGAN_ACT_TANH) v = tanhf(v);
      const int P = S * q + r;
      int ii = P - a.o_padL;
      const size_t rowoff = (size_t)b * N + n;
      if (ii >= 0 && ii < a.Lout) {
        if (a.o_roll != 0) {
          ii -= a.o_roll;
          if (ii < 0) ii += a.Lout;
          if (ii >= a.Lout) ii -= a.Lout;
        }
        a.out0[rowoff * (size_t)a.Lout + ii] = v;
      } else if (a.halo != nullptr) {
        const int hl = a.o_padL + a.o_padR;
        if (ii < 0) a.halo[rowoff * hl + P] = v;
        else if (ii - a.Lout < a.o_padR) a.halo[rowoff * hl + a.o_padL + (ii - a.Lout)] = v;
      }
    }
}

template <int S, int N, int KT>
static int launch_tsmall_snk(const CorrArgs& a, const float* w, int K, int M, int pad,
                             hipStream_t st) {
  dim3 grid(ceil_div(a.Tcols, 256), a.B);
  switch (pad % S) {
    case 0: hipLaunchKernelGGL((tsmall_kernel<S, N, 0, KT>), grid, dim3(256), 0, st, a, w, K, M); break;
    case 1: hipLaunchKernelGGL((tsmall_kernel<S, N, 1 % S, KT>), grid, dim3(256), 0, st, a, w, K, M); break;
    case 2: hipLaunchKernelGGL((tsmall_kernel<S, N, 2 % S, KT>), grid, dim3(256), 0, st, a, w, K, M); break;
    default: hipLaunchKernelGGL((tsmall_kernel<S, N, 3 % S, KT>), grid, dim3(256), 0, st, a, w, K, M); break;
  }
  return segan_check_launch("tsmall_kernel");
}

template <int S, int N>
static int launch_tsmall_sn(const CorrArgs& a, const float* w, int K, int M, int pad,
                            hipStream_t st) {
  if (K == 31) return launch_tsmall_snk<S, N, 31>(a, w, K, M, pad, st);
  return launch_tsmall_snk<S, N, 0>(a, w, K, M, pad, st);
}


// The same T form with FOUR consecutive low-rate positions per thread (stride 4, width 31:
// the SEGAN geometry).  The one-position kernel above is LDS-issue bound (9 window reads per
// 31 FMAs); here a thread reads its 12-entry window with three 16-byte LDS loads, does 4 x 31
// FMAs per output channel on it, and owns 16 consecutive output samples (four 16-byte stores
// when the run is aligned and unrolled).  A workgroup covers 1024 positions of one sample.
template <int N, int PM>
__global__ __launch_bounds__(256) void tsmall4_kernel(const CorrArgs a, const float* __restrict__ w,
                                                      int M) {
  constexpr int S = 4, U = 8, KT = 31, Q = 4;
  constexpr int MC = 8;                    // input channels per LDS chunk
  constexpr int TWQ = Q * 256 + 12;        // window entries a workgroup touches (>= 1024 + U + 1)
  __shared__ __attribute__((aligned(16))) float xs[MC][TWQ];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * (Q * 256);
  float acc[Q][S][N];
#pragma unroll
  for (int i = 0; i < Q; ++i)
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
      for (int n = 0; n < N; ++n) acc[i][r][n] = 0.0f;
  // staging: window entry j <-> input time q0 + win_start + j; thread owns entries
  // tid + 256*i (i < 4) and 1024 + tid (tid < 12)
  int so[5];
  unsigned sok = 0u;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = tid + 256 * i;
    const int t = q0 + a.win_start + j;
    const bool ok = (i < 4 || tid < 12) && t >= 0 && t < a.Lin;
    so[i] = ok ? t : 0;
    if (ok) sok |= 1u << i;
  }
  const int bo0 = b * a.in.C0 * a.Lin, bo1 = b * a.in.C1 * a.Lin;
  // the 8 x 5 loads of the NEXT chunk of input channels are issued before the FMAs of the current one
  // (round 6: issued after them, every chunk began with an exposed HBM round trip)
  float pv[MC][5];
  auto load_rows = [&](int mc0) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      const int m = mc0 + mc < M ? mc0 + mc : 0;
      const bool seg1 = m >= a.in.C0;
      const float* rowp = seg1 ? a.in.p1 + (size_t)(m - a.in.C0) * a.Lin + bo1
                               : a.in.p0 + (size_t)m * a.Lin + bo0;
#pragma unroll
      for (int i = 0; i < 5; ++i) pv[mc][i] = rowp[so[i]];
    }
  };
  load_rows(0);
  for (int mc0 = 0; mc0 < M; mc0 += MC) {
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      const bool mok = mc0 + mc < M;
      const ChanXf xf = segan_chan_xf(a.in, mok ? mc0 + mc : 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xs[mc][tid + 256 * i] = (mok && ((sok >> i) & 1u)) ? segan_apply_xf(xf, pv[mc][i]) : 0.0f;
      if (tid < 12) xs[mc][1024 + tid] = (mok && ((sok >> 4) & 1u)) ? segan_apply_xf(xf, pv[mc][4]) : 0.0f;
    }
    __syncthreads();
    if (mc0 + MC < M) load_rows(mc0 + MC);
    const int mcn = min(MC, M - mc0);
    for (int mc = 0; mc < mcn; ++mc) {
      float xv[12];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&xs[mc][4 * tid + 4 * i]);
        xv[4 * i] = t4[0]; xv[4 * i + 1] = t4[1]; xv[4 * i + 2] = t4[2]; xv[4 * i + 3] = t4[3];
      }
      const float* wm = w + (size_t)(mc0 + mc) * N * KT;
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const int rho = (r + PM) % S, cs = (r + PM) / S;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = S * u + rho;
          if (k < KT) {
#pragma unroll
            for (int n = 0; n < N; ++n) {
              const float wv = wm[n * KT + k];
#pragma unroll
              for (int i = 0; i < Q; ++i)
                acc[i][r][n] = fmaf(wv, xv[i + cs + (U - 1) - u], acc[i][r][n]);
            }
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const float bs = a.bias ? a.bias[n] : 0.0f;
    const size_t rowoff = (size_t)b * N + n;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int q = q0 + Q * tid + i;
      if (q >= a.Tcols) continue;
      float v[S];
#pragma unroll
      for (int r = 0; r < S; ++r) {
        v[r] = acc[i][r][n] + bs;
        if (a.act == SEGAN_ACT_TANH) v[r] = tanhf(v[r]);
      }
      const int i0 = S * q - a.o_padL;
      if (a.o_roll == 0 && i0 >= 0 && i0 + 3 < a.Lout && (a.o_padL & 3) == 0) {
        const f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(a.out0 + rowoff * (size_t)a.Lout + i0) = o;
        continue;
      }
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const int P = S * q + r;
        int ii = P - a.o_padL;
        if (ii >= 0 && ii < a.Lout) {
          if (a.o_roll != 0) {
            ii -= a.o_roll;
            if (ii < 0) ii += a.Lout;
            if (ii >= a.Lout) ii -= a.Lout;
          }
          a.out0[rowoff * (size_t)a.Lout + ii] = v[r];
        } else if (a.halo != nullptr) {
          const int hl = a.o_padL + a.o_padR;
          if (ii < 0) a.halo[rowoff * hl + P] = v[r];
          else if (ii - a.Lout < a.o_padR) a.halo[rowoff * hl + a.o_padL + (ii - a.Lout)] = v[r];
        }
      }
    }
  }
}

template <int N>
static int launch_tsmall4(const CorrArgs& a, const float* w, int M, int pad, hipStream_t st) {
  dim3 grid(ceil_div(a.Tcols, 1024), a.B);
  switch (pad % 4) {
    case 0: hipLaunchKernelGGL((tsmall4_kernel<N, 0>), grid, dim3(256), 0, st, a, w, M); break;
    case 1: hipLaunchKernelGGL((tsmall4_kernel<N, 1>), grid, dim3(256), 0, st, a, w, M); break;
    case 2: hipLaunchKernelGGL((tsmall4_kernel<N, 2>), grid, dim3(256), 0, st, a, w, M); break;
    default: hipLaunchKernelGGL((tsmall4_kernel<N, 3>), grid, dim3(256), 0, st, a, w, M); break;
  }
  return segan_check_launch("tsmall4_kernel");
}

// `a` is filled exactly as for the MFMA T form; w is the UNPACKED weight [M][N][K]
int segan_launch_tsmall(CorrArgs& a, const float* w, int K, int M, int N, int S, int pad,
                         hipStream_t st) {
  if (int e = segan_src_defaults(&a.in, st, "tsmall")) return e;
  if (S == 4 && K == 31 && a.Tcols >= 1024)
    return N == 1 ? launch_tsmall4<1>(a, w, M, pad, st) : launch_tsmall4<2>(a, w, M, pad, st);
  if (N == 1) {
    if (S == 4) return launch_tsmall_sn<4, 1>(a, w, K, M, pad, st);
    if (S == 2) return launch_tsmall_sn<2, 1>(a, w, K, M, pad, st);
    return launch_tsmall_sn<1, 1>(a, w, K, M, pad, st);
  }
  if (S == 4) return launch_tsmall_sn<4, 2>(a, w, K, M, pad, st);
  if (S == 2) return launch_tsmall_sn<2, 2>(a, w, K, M, pad, st);
  return launch_tsmall_sn<1, 2>(a, w, K, M, pad, st);
}


// Staging of the F-form edge kernels with a thread's loads IN FLIGHT TOGETHER (round 6): written as
// rolled loops `for (j = tid; j < n; j += 256) lds[j] = f(global[g(j)])` the compiler issues load -
// s_waitcnt vmcnt(0) - store once per element, i.e. 17 (+ 16 for the weights) dependent HBM round
// trips per workgroup before its first FMA — most of the kernel's duration.  Addresses are clamped so
// that the loads are unconditional; the mask is applied to the loaded value.
template <int S, int N, int XW>
__device__ __forceinline__ void fsmall_stage_window(const CorrArgs& a, int b, int t0, int tid,
                                                    float (&xs)[N][XW]) {
  constexpr int NX = (XW + 255) / 256;
  constexpr int BT = 6;                      // positions per batch (x N channels of loads in flight)
  const float* rows[N];
  ChanXf xf[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    rows[n] = segan_src_row(a.in, b, n, a.Lin);
    xf[n] = segan_chan_xf(a.in, n);
  }
#pragma unroll
  for (int g0 = 0; g0 < NX; g0 += BT) {
    float v[N][BT];
    int idx[BT];
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int j = tid + 256 * (g0 + i);
      idx[i] = (g0 + i < NX && j < XW) ? segan_hi_index(S * t0 + j, a.Lin, a.padL, a.mode, a.roll) : -1;
#pragma unroll
      for (int n = 0; n < N; ++n) v[n][i] = rows[n][idx[i] >= 0 ? idx[i] : 0];
    }
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int j = tid + 256 * (g0 + i);
      if (g0 + i < NX && j < XW) {
#pragma unroll
        for (int n = 0; n < N; ++n) xs[n][j] = idx[i] >= 0 ? segan_apply_xf(xf[n], v[n][i]) : 0.0f;
      }
    }
  }
}

// 64 output channels x N x 32 taps of the packed F layout into ws[ml * WST + n * 32 + k]
template <int S, int N, int WST>
__device__ __forceinline__ void fsmall_stage_weights(const CorrArgs& a, int m0, int tid, float* ws) {
  constexpr int U = 32 / S;
  constexpr int NW = 64 * N * 32 / 256;      // elements per thread
  constexpr int BT = 8;
  static_assert(NW % BT == 0, "weight staging batches");
#pragma unroll
  for (int g0 = 0; g0 < NW; g0 += BT) {
    float v[BT];
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int e = tid + 256 * (g0 + i);
      const int ml = e & 63, nk = e >> 6;
      const int n = nk >> 5, k = nk & 31;
      const int row = (n * S + k % S) * U + k / S;
      const int m = m0 + ml;
      v[i] = a.wp[(size_t)row * a.RP + (m < a.RP ? m : 0)];
      v[i] = m < a.RP ? v[i] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int e = tid + 256 * (g0 + i);
      const int ml = e & 63, nk = e >> 6;
      ws[ml * WST + (nk >> 5) * 32 + (nk & 31)] = v[i];
    }
  }
}

// ====================================================================================
// F form for 1-2 input channels (the first conv of G and of D: HBM-bound, and an MFMA tile
// whose contraction is N*32 <= 64 deep would be mostly the padding to the 64-deep LDS chunk).
// Direct VALU kernel: a workgroup owns 256 output positions of one sample, stages the padded
// input window once (reflect / roll / transform applied while staging) and walks the output
// channels with the taps read as 16-byte LDS broadcasts from a zero-padded [m][n][32] copy of
// the packed weights.  Stores are coalesced along time.
// ====================================================================================
template <int S, int N>
__global__ __launch_bounds__(256) void fsmall_kernel(const CorrArgs a, int M) {
  constexpr int XW = S * 256 + 32;
  constexpr int WST = N * 32 + 4;          // row stride of the weight copy (16-B aligned)
  __shared__ __attribute__((aligned(16))) float xs[N][XW];
  __shared__ __attribute__((aligned(16))) float ws[64 * WST];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 256;
  fsmall_stage_window<S, N, XW>(a, b, t0, tid, xs);
  const int t = t0 + tid;
  for (int m0 = 0; m0 < M; m0 += 64) {
    // packed F layout: w[m][n][S*u + r] = wp[((n*S + r)*U + u) * RP + m]; rows of taps >= K
    // are zero.  Lanes run along m so the global reads are coalesced.
    fsmall_stage_weights<S, N, WST>(a, m0, tid, ws);
    __syncthreads();
    float xv[N][32];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if (S == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&xs[n][4 * tid + 4 * i]);
          xv[n][4 * i] = v[0]; xv[n][4 * i + 1] = v[1]; xv[n][4 * i + 2] = v[2]; xv[n][4 * i + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) xv[n][k] = xs[n][S * tid + k];
      }
    }
    const int mcn = min(64, M - m0);
    for (int ml = 0; ml < mcn; ++ml) {
      float acc = a.bias ? a.bias[m0 + ml] : 0.0f;
#pragma unroll
      for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&ws[ml * WST + n * 32 + 4 * i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = fmaf(wv[e], xv[n][4 * i + e], acc);
        }
      }
      if (t < a.Lout) a.out0[((size_t)b * M + m0 + ml) * a.Lout + t] = acc;
    }
    __syncthreads();
  }
}


// The same F form with FOUR consecutive output positions per thread (stride 4): the per-m tap
// broadcast (16 LDS reads for 2 channels) then feeds 4 x 62 FMAs instead of 62, and a thread
// stores 16 contiguous bytes per output channel.  A workgroup covers 1024 positions.
template <int N>
__global__ __launch_bounds__(256) void fsmall4_kernel(const CorrArgs a, int M) {
  constexpr int S = 4, Q = 4;
  constexpr int XW = S * Q * 256 + 32;     // padded input samples a workgroup touches
  constexpr int WST = N * 32 + 4;
  constexpr int XR = S * (Q - 1) + 32;     // 44 input samples per thread and channel
  __shared__ __attribute__((aligned(16))) float xs[N][XW];
  __shared__ __attribute__((aligned(16))) float ws[64 * WST];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * (Q * 256);
  fsmall_stage_window<S, N, XW>(a, b, t0, tid, xs);
  const int t = t0 + Q * tid;
  for (int m0 = 0; m0 < M; m0 += 64) {
    fsmall_stage_weights<S, N, WST>(a, m0, tid, ws);
    __syncthreads();
    float xv[N][XR];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int i = 0; i < XR / 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&xs[n][S * Q * tid + 4 * i]);
        xv[n][4 * i] = v[0]; xv[n][4 * i + 1] = v[1]; xv[n][4 * i + 2] = v[2]; xv[n][4 * i + 3] = v[3];
      }
    const int mcn = min(64, M - m0);
    for (int ml = 0; ml < mcn; ++ml) {
      const float bs = a.bias ? a.bias[m0 + ml] : 0.0f;
      float acc[Q] = {bs, bs, bs, bs};
#pragma unroll
      for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&ws[ml * WST + n * 32 + 4 * i]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = fmaf(wv[e], xv[n][S * q + 4 * i + e], acc[q]);
        }
      }
      float* o = a.out0 + ((size_t)b * M + m0 + ml) * a.Lout + t;
      if (t + Q <= a.Lout && (a.Lout & 3) == 0) {
        const f32x4 ov = {acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(o) = ov;
      } else {
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (t + q < a.Lout) o[q] = acc[q];
      }
    }
    __syncthreads();
  }
}

int segan_launch_fsmall(CorrArgs& a, int M, int N, int S, hipStream_t st) {
  if (int e = segan_src_defaults(&a.in, st, "fsmall")) return e;
  if (S == 4 && a.Lout >= 1024) {
    dim3 grid4(ceil_div(a.Lout, 1024), a.B);
    if (N == 1) hipLaunchKernelGGL((fsmall4_kernel<1>), grid4, dim3(256), 0, st, a, M);
    else hipLaunchKernelGGL((fsmall4_kernel<2>), grid4, dim3(256), 0, st, a, M);
    return segan_check_launch("fsmall4_kernel");
  }
  dim3 grid(ceil_div(a.Lout, 256), a.B);
#define FS(SS, NN) hipLaunchKernelGGL((fsmall_kernel<SS, NN>), grid, dim3(256), 0, st, a, M)
  if (N == 1) { if (S == 4) FS(4, 1); else if (S == 2) FS(2, 1); else FS(1, 1); }
  else { if (S == 4) FS(4, 2); else if (S == 2) FS(2, 2); else FS(1, 2); }
#undef FS
  return segan_check_launch("fsmall_kernel");
}
