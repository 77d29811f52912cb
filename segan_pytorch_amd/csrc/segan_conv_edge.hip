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
  // accumulators of two ADJACENT positions share a register pair (i = 2*ip, 2*ip + 1): see the FMAs
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[Q / 2][S][N];
#pragma unroll
  for (int i = 0; i < Q / 2; ++i)
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
      for (int n = 0; n < N; ++n) acc[i][r][n] = f32x2{0.0f, 0.0f};
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
      // PACKED FMAs over pairs of positions (round 6; scripts/micro/pkfma.hip: 138 TF/s against 75 for
      // v_fmac_f32 — the kernel ran at 45 - 52 TF/s): the tap is one half of an SGPR pair broadcast to
      // both lanes by op_sel, the samples x[j], x[j+1] a VGPR pair — the 12-sample window is held
      // twice, as the even pairs (0,1) .. (10,11) of the 16-byte reads and as the odd pairs (1,2) ..
      // (9,10) read again from LDS.  Every accumulator sees the same FMA sequence as before.
      f32x2 xe[6], xo[5];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&xs[mc][4 * tid + 4 * i]);
        xe[2 * i] = f32x2{t4[0], t4[1]};
        xe[2 * i + 1] = f32x2{t4[2], t4[3]};
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) xo[i] = f32x2{xs[mc][4 * tid + 2 * i + 1], xs[mc][4 * tid + 2 * i + 2]};
      // the N x 31 taps of input channel m as pairs of the flat index f = n*31 + k (wave-uniform
      // scalar loads, 4-byte aligned); with N = 1 the odd tap 30 is the high half of the pair (29, 30)
      typedef const __attribute__((address_space(4))) float cfl;
      constexpr int NPW = (N * KT + 1) / 2;
      cfl* wm = (cfl*)w + (size_t)(mc0 + mc) * N * KT;
      f32x2 wp[NPW];
#pragma unroll
      for (int p = 0; p < NPW; ++p) {
        const int f0 = 2 * p + 1 < N * KT ? 2 * p : N * KT - 2;
        wp[p] = f32x2{wm[f0], wm[f0 + 1]};
      }
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const int rho = (r + PM) % S, cs = (r + PM) / S;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = S * u + rho;
          if (k < KT) {
#pragma unroll
            for (int n = 0; n < N; ++n) {
              const int f = n * KT + k;
              const bool last = 2 * (f >> 1) + 1 >= N * KT;     // the unpaired last tap (N*KT odd)
              const int p = f >> 1, h = last ? 1 : (f & 1);
#pragma unroll
              for (int ip = 0; ip < Q / 2; ++ip) {
                const int j = 2 * ip + cs + (U - 1) - u;
                const f32x2 xp = (j & 1) ? xo[j >> 1] : xe[j >> 1];
                if (h)
                  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                      : "+v"(acc[ip][r][n]) : "s"(wp[p]), "v"(xp));
                else
                  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
                      : "+v"(acc[ip][r][n]) : "s"(wp[p]), "v"(xp));
              }
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
        v[r] = acc[i >> 1][r][n][i & 1] + bs;
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


// The same F form with FOUR consecutive output positions per thread (strides 4 and 2), the taps as
// WAVE-UNIFORM SCALAR LOADS and the FMAs PACKED over pairs of output channels (round 6).  The layer is
// VALU-bound: 62 FMAs per output on 1 - 2 input channels, 9.75 GFLOP per D call; scripts/micro/pkfma.hip
// measures the issue peaks on this chip as 75 TF/s for v_fmac_f32 and 138 TF/s for v_pk_fma_f32 with an
// SGPR-pair operand, and the former version (taps as 16-byte LDS broadcasts, v_fmac) ran at 57 TF/s.
// A pass owns MC = 8 output channels; for every (n, k) the eight channels' taps are 32 contiguous bytes
// of the packed F layout — one s_load_dwordx8 through the scalar cache — and feed 4 pairs x 4 positions
// of v_pk_fma_f32: the two channels' taps are the instruction's SGPR pair, the input sample one half
// of a VGPR pair broadcast to both lanes by op_sel.  A workgroup covers 1024 positions; LDS holds the
// input window only.
// RPT: the pitch of the packed weights when known at compile time (64: every SEGAN first layer has
// <= 64 output channels; the 64 row offsets of a pass are then immediates of the s_loads instead of 64
// loop-invariant products the compiler keeps live in SGPRs and spills) or 0 = runtime pitch.
template <int S, int N, int RPT>
__global__ __launch_bounds__(256) void fsmall4_kernel(const CorrArgs a, const float* __restrict__ wq,
                                                      const float* __restrict__ bias, int M) {
  static_assert(S == 2 || S == 4, "fsmall4: strides 2 and 4");
  constexpr int Q = 4, U = 32 / S, MC = 8, CP = MC / 2;
  constexpr int XW = S * Q * 256 + 32;     // padded input samples a workgroup touches
  constexpr int XR = (S * (Q - 1) + 32 + 3) / 4 * 4;     // 44 (40) input samples per thread and channel
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float xs[N][XW];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * (Q * 256);
  fsmall_stage_window<S, N, XW>(a, b, t0, tid, xs);
  __syncthreads();
  const int t = t0 + Q * tid;
  f32x2 xv[N][XR / 2];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int i = 0; i < XR / 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&xs[n][S * Q * tid + 4 * i]);
      xv[n][2 * i] = f32x2{v[0], v[1]};
      xv[n][2 * i + 1] = f32x2{v[2], v[3]};
    }
  const int RP = RPT ? RPT : a.RP;
  const bool vec = (a.Lout & 3) == 0 && t + Q <= a.Lout;
  // constant address space: a wave-uniform load from it is always an s_load (the global stores of the
  // previous pass otherwise make the compiler fall back to per-lane global loads)
  typedef const __attribute__((address_space(4))) f32x2 cf2;
  // packed F layout: w[m][n][S*u + r] = wq[((n*S + r)*U + u) * RP + m]; rows of taps >= K and the
  // columns M <= m < RP (a multiple of 64) hold zeros, so a pass may read past M
  for (int m0 = 0; m0 < M; m0 += MC) {
    f32x2 acc[CP][Q];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const int ma = m0 + 2 * c < M ? m0 + 2 * c : M - 1, mb = m0 + 2 * c + 1 < M ? m0 + 2 * c + 1 : M - 1;
      const f32x2 bs = bias ? f32x2{bias[ma], bias[mb]} : f32x2{0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[c][q] = bs;
    }
    // a GROUP = four consecutive taps k = 4g .. 4g+3 of one input channel (four s_load_dwordx8, 32 SGPRs);
    // two groups' registers alternate.  Scalar loads return out of order, so a wait for one group is a
    // wait for everything outstanding: the next group's loads are issued right AFTER the wait for the
    // current one (its first tap's FMAs) and fly during the remaining three taps' FMAs.  The scheduling
    // barriers keep the compiler from hoisting all 8N groups to the top (it then spills SGPRs into
    // VGPR lanes); the FMA is spelled in assembly to pin the operand classes (left alone, the SLP
    // vectoriser packs the channels too, but on splatted copies of the 44 x N window registers).
    f32x2 wa[4][CP], wb[4][CP];
    auto ld = [&](f32x2 (&w)[4][CP], int g) {
      const int n = g >> 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 4 * (g & 7) + e;     // tap k = S*u + r
        cf2* wr = (cf2*)(wq + m0) + (((n * S + k % S) * U + k / S) * RP >> 1);
#pragma unroll
        for (int c = 0; c < CP; ++c) w[e][c] = wr[c];
      }
    };
    auto fm = [&](const f32x2 (&w)[4][CP], int g, int e0, int e1) {
      const int n = g >> 3;
#pragma unroll
      for (int e = e0; e < e1; ++e)
#pragma unroll
        for (int c = 0; c < CP; ++c)
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const int j = S * q + 4 * (g & 7) + e;     // sample x[S*q + k]: half (j & 1) of pair j / 2
            if (j & 1)
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]"
                  : "+v"(acc[c][q]) : "s"(w[e][c]), "v"(xv[n][j >> 1]));
            else
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]"
                  : "+v"(acc[c][q]) : "s"(w[e][c]), "v"(xv[n][j >> 1]));
          }
    };
    ld(wa, 0);
#pragma unroll
    for (int g = 0; g < 8 * N; g += 2) {
      fm(wa, g, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      ld(wb, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      fm(wa, g, 1, 4);
      fm(wb, g + 1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < 8 * N) ld(wa, g + 2);
      __builtin_amdgcn_sched_barrier(0);
      fm(wb, g + 1, 1, 4);
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      if (m0 + c >= M) break;
      float* o = a.out0 + ((size_t)b * M + m0 + c) * a.Lout + t;
      if (vec) {
        const f32x4 ov = {acc[c / 2][0][c & 1], acc[c / 2][1][c & 1], acc[c / 2][2][c & 1], acc[c / 2][3][c & 1]};
        *reinterpret_cast<f32x4*>(o) = ov;
      } else {
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (t + q < a.Lout) o[q] = acc[c / 2][q][c & 1];
      }
    }
  }
}

int segan_launch_fsmall(CorrArgs& a, int M, int N, int S, hipStream_t st) {
  if (int e = segan_src_defaults(&a.in, st, "fsmall")) return e;
  if ((S == 4 || S == 2) && a.Lout >= 1024) {
    dim3 grid4(ceil_div(a.Lout, 1024), a.B);
#define FS4(SS, NN, RR) hipLaunchKernelGGL((fsmall4_kernel<SS, NN, RR>), grid4, dim3(256), 0, st, a, a.wp, a.bias, M)
    if (S == 4) {
      if (a.RP == 64) { if (N == 1) FS4(4, 1, 64); else FS4(4, 2, 64); }
      else { if (N == 1) FS4(4, 1, 0); else FS4(4, 2, 0); }
    } else {
      if (a.RP == 64) { if (N == 1) FS4(2, 1, 64); else FS4(2, 2, 64); }
      else { if (N == 1) FS4(2, 1, 0); else FS4(2, 2, 0); }
    }
#undef FS4
    return segan_check_launch("fsmall4_kernel");
  }
  dim3 grid(ceil_div(a.Lout, 256), a.B);
#define FS(SS, NN) hipLaunchKernelGGL((fsmall_kernel<SS, NN>), grid, dim3(256), 0, st, a, M)
  if (N == 1) { if (S == 4) FS(4, 1); else if (S == 2) FS(2, 1); else FS(1, 1); }
  else { if (S == 4) FS(4, 2); else if (S == 2) FS(2, 2); else FS(1, 2); }
#undef FS
  return segan_check_launch("fsmall_kernel");
}
