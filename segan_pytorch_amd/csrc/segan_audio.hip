// segan_audio.hip — the audio-side pieces next to the GAN step:
//   * de-emphasis  x[n] = coef * x[n-1] + y[n]   (segan/datasets/se_dataset.py:119-126, a
//     per-sample python loop in the reference) as a blocked parallel scan;
//   * segmental SNR (segan/utils.py:350-395, numpy) for on-device validation.
#include "segan_common.h"

// ---------------------------------------------------------------------------------
// De-emphasis.  One workgroup per row; a row is walked in slabs of 1024 threads x DE_E
// samples.  A thread runs the recurrence over its DE_E samples from a zero state; the pairs
// (decay a = coef^n, response b) of the threads are combined by a block scan with
//   (a2, b2) o (a1, b1) = (a2*a1, a2*b1 + b2)
// and the carry of the row is threaded through the slabs.  Arithmetic in double (the reference
// accumulates in float64 under numpy < 2 and in float32 under numpy 2: both are within 2e-6
// of this), rounded to fp32 once per output sample.
// ---------------------------------------------------------------------------------
#define DE_T 1024
#define DE_E 8

__global__ __launch_bounds__(DE_T) void deemphasis_kernel(const float* __restrict__ y,
                                                          float* __restrict__ x, int T,
                                                          double coef) {
  __shared__ double sa[DE_T], sb[DE_T];
  __shared__ double s_carry;
  const int t = threadIdx.x;
  const float* yr = y + (size_t)blockIdx.x * T;
  float* xr = x + (size_t)blockIdx.x * T;
  if (t == 0) s_carry = 0.0;
  __syncthreads();
  double cpow[DE_E + 1];
  cpow[0] = 1.0;
#pragma unroll
  for (int e = 1; e <= DE_E; ++e) cpow[e] = cpow[e - 1] * coef;
  for (int base = 0; base < T; base += DE_T * DE_E) {
    const int i0 = base + t * DE_E;
    double loc[DE_E];
    double b = 0.0;
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < DE_E; ++e) {
      if (i0 + e < T) {
        b = coef * b + (double)yr[i0 + e];
        ++cnt;
      }
      loc[e] = b;
    }
    double a = cpow[cnt];
    sa[t] = a;
    sb[t] = b;
    __syncthreads();
    // inclusive Hillis-Steele scan of the (a, b) pairs
    for (int off = 1; off < DE_T; off <<= 1) {
      double pa = 1.0, pb = 0.0;
      if (t >= off) { pa = sa[t - off]; pb = sb[t - off]; }
      __syncthreads();
      if (t >= off) {
        sb[t] = sa[t] * pb + sb[t];
        sa[t] = sa[t] * pa;
      }
      __syncthreads();
    }
    // state entering this thread's samples: exclusive prefix applied to the row carry
    const double carry = s_carry;
    const double cin = t == 0 ? carry : sa[t - 1] * carry + sb[t - 1];
#pragma unroll
    for (int e = 0; e < DE_E; ++e)
      if (i0 + e < T) xr[i0 + e] = (float)(loc[e] + cpow[e + 1] * cin);
    __syncthreads();
    if (t == DE_T - 1) s_carry = sa[t] * carry + sb[t];
    __syncthreads();
  }
}

extern "C" int segan_deemphasis(const float* y, float* x, int rows, int T, double coef,
                                void* stream) {
  SEGAN_REQUIRE(y && x, "deemphasis: NULL pointer");
  SEGAN_REQUIRE(rows > 0 && T > 0, "deemphasis: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (coef <= 0.0) {     // se_dataset.py:120-121: returned unchanged
    if (x != y && hipMemcpyAsync(x, y, (size_t)rows * T * sizeof(float), hipMemcpyDeviceToDevice,
                                 st) != hipSuccess) {
      segan_set_error("deemphasis: copy failed");
      return SEGAN_ELAUNCH;
    }
    return SEGAN_OK;
  }
  hipLaunchKernelGGL(deemphasis_kernel, dim3(rows), dim3(DE_T), 0, st, y, x, T, coef);
  return segan_check_launch("deemphasis_kernel");
}

// ---------------------------------------------------------------------------------
// Segmental SNR, utils.py:350-395: 30 ms frames (winlength = round(30*srate/1000)), hop
// winlength/4, window 0.5*(1 - cos(2*pi*k/(winlength+1))), k = 1..winlength; per frame
// 10*log10(E_clean / (E_noise + eps) + eps) clamped to [-10, 35]; plus the overall SNR
// 10*log10(sum ref^2 / (sum (ref-deg)^2 + 10e-20)).  float64 like numpy (the window is).
// One wave per frame.
// ---------------------------------------------------------------------------------
extern "C" int segan_ssnr_frames(int T, int srate) {
  if (T <= 0 || srate <= 0) return 0;
  const int win = (int)__builtin_round(30.0 * srate / 1000.0);
  const int skip = win / 4;
  if (skip <= 0) return 0;
  const int nf = (int)((double)T / skip - (double)win / skip);
  return nf > 0 ? nf : 0;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void ssnr_frames_kernel(const float* __restrict__ ref,
                                                          const float* __restrict__ deg,
                                                          float* __restrict__ seg, int T,
                                                          int nframes, int win, int skip,
                                                          double eps) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= nframes) return;
  const float* r = ref + (size_t)blockIdx.y * T + (size_t)f * skip;
  const float* d = deg + (size_t)blockIdx.y * T + (size_t)f * skip;
  double es = 0.0, en = 0.0;
  for (int k = lane; k < win; k += 64) {
    const double w = 0.5 * (1.0 - cos(2.0 * 3.14159265358979323846 * (double)(k + 1) / (double)(win + 1)));
    const double c = (double)r[k] * w, p = (double)d[k] * w;
    es += c * c;
    en += (c - p) * (c - p);
  }
  es = wave_sum_d(es);
  en = wave_sum_d(en);
  if (lane == 0) {
    double s = 10.0 * log10(es / (en + eps) + eps);
    s = s < -10.0 ? -10.0 : s;
    s = s > 35.0 ? 35.0 : s;
    seg[(size_t)blockIdx.y * nframes + f] = (float)s;
  }
}

__global__ __launch_bounds__(1024) void ssnr_overall_kernel(const float* __restrict__ ref,
                                                            const float* __restrict__ deg,
                                                            const float* __restrict__ seg,
                                                            float* __restrict__ out, int T,
                                                            int nframes) {
  __shared__ double s0[16], s1[16], s2[16];
  const float* r = ref + (size_t)blockIdx.x * T;
  const float* d = deg + (size_t)blockIdx.x * T;
  double a = 0.0, b = 0.0, m = 0.0;
  for (int i = threadIdx.x; i < T; i += 1024) {
    const double x = r[i], e = (double)r[i] - (double)d[i];
    a += x * x;
    b += e * e;
  }
  for (int i = threadIdx.x; i < nframes; i += 1024) m += seg[(size_t)blockIdx.x * nframes + i];
  a = wave_sum_d(a); b = wave_sum_d(b); m = wave_sum_d(m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { s0[w] = a; s1[w] = b; s2[w] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double A = 0, B = 0, M = 0;
    for (int k = 0; k < 16; ++k) { A += s0[k]; B += s1[k]; M += s2[k]; }
    out[2 * blockIdx.x] = (float)(10.0 * log10(A / (B + 10e-20)));
    out[2 * blockIdx.x + 1] = nframes > 0 ? (float)(M / nframes) : 0.0f;
  }
}

extern "C" int segan_ssnr(const float* ref, const float* deg, float* seg, float* out, int rows,
                          int T, int srate, double eps, void* stream) {
  SEGAN_REQUIRE(ref && deg && seg && out, "ssnr: NULL pointer");
  SEGAN_REQUIRE(rows > 0 && T > 0 && srate > 0, "ssnr: bad sizes");
  const int win = (int)__builtin_round(30.0 * srate / 1000.0);
  const int skip = win / 4;
  const int nf = segan_ssnr_frames(T, srate);
  hipStream_t st = (hipStream_t)stream;
  if (nf > 0) {
    hipLaunchKernelGGL(ssnr_frames_kernel, dim3(ceil_div(nf, 4), rows), dim3(256), 0, st, ref, deg,
                       seg, T, nf, win, skip, eps);
    if (int e = segan_check_launch("ssnr_frames_kernel")) return e;
  }
  hipLaunchKernelGGL(ssnr_overall_kernel, dim3(rows), dim3(1024), 0, st, ref, deg, seg, out, T, nf);
  return segan_check_launch("ssnr_overall_kernel");
}
