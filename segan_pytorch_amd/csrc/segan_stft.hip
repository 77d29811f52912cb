// segan_stft.hip — the STFT power loss of the WSEGAN step (model.py:640-653):
//   torch.stft(x, n_fft, hop_length=160, win_length=320, normalized=True)  (window=None)
//   10*log10(|X|^2 + 10e-20), L1 between the enhanced and the clean spectra.
// torch pads the (rectangular, all-ones) 320-sample window to n_fft with zeros on both sides
// and reflect-pads the signal by n_fft/2 (center=True), so a frame has only `win` non-zero
// samples and the transform is a dense [win] x [2*(n_fft/2+1)] real matrix product:
//   frames[b*NF+f][j] = xpad[b][f*hop + left + j],            left = (n_fft - win)/2
//   S = frames x basis,  basis[j][k] = cos(2 pi k (left+j)/n_fft)/sqrt(n_fft),
//                        basis[j][nbins+k] = -sin(...)/sqrt(n_fft)
// The product itself is segan_gemm (exact fp32 MFMA); this file holds the framing, the
// power/log stage, their backward, and the overlap-add that returns the frame gradient to
// the waveform.
#include "segan_common.h"
#include <math.h>

static __device__ __forceinline__ int stft_reflect(int o, int T) {
  if (o < 0) o = -o;
  if (o >= T) o = 2 * (T - 1) - o;
  return o;
}

__global__ void stft_basis_kernel(float* __restrict__ basis, int n_fft, int win, int left, int nbins,
                                  int pitch) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= win * nbins) return;
  const int j = idx / nbins, k = idx - j * nbins;
  if (k < pitch - 2 * nbins) basis[(size_t)j * pitch + 2 * nbins + k] = 0.0f;   // pad columns
  // exact argument reduction in integers, then double precision sincospi
  const long m = ((long)k * (left + j)) % n_fft;
  double s, c;
  sincospi(2.0 * (double)m / (double)n_fft, &s, &c);
  const double nrm = 1.0 / sqrt((double)n_fft);
  basis[(size_t)j * pitch + k] = (float)(c * nrm);
  basis[(size_t)j * pitch + nbins + k] = (float)(-s * nrm);
}

extern "C" int segan_stft_pitch(int n_fft) { return n_fft >= 2 ? round_up(2 * (n_fft / 2 + 1), 4) : 0; }

extern "C" int segan_stft_basis(float* basis, int n_fft, int win, void* stream) {
  SEGAN_REQUIRE(basis, "stft_basis: NULL pointer");
  SEGAN_REQUIRE(n_fft >= 2 && win >= 1 && win <= n_fft, "stft_basis: need 1 <= win <= n_fft");
  const int nbins = n_fft / 2 + 1;
  const int total = win * nbins;
  hipLaunchKernelGGL(stft_basis_kernel, dim3(ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, basis, n_fft, win, (n_fft - win) / 2, nbins,
                     segan_stft_pitch(n_fft));
  return segan_check_launch("stft_basis");
}

__global__ void stft_frames_kernel(const float* __restrict__ x, float* __restrict__ frames, int B,
                                   int T, int NF, int hop, int win, int off) {
  const size_t total = (size_t)B * NF * win;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % win);
    const size_t bf = i / win;
    const int f = (int)(bf % NF);
    const int b = (int)(bf / NF);
    frames[i] = x[(size_t)b * T + stft_reflect(f * hop + j + off, T)];
  }
}

extern "C" int segan_stft_frames(const float* x, float* frames, int B, int T, int n_fft, int hop,
                                 int win, void* stream) {
  SEGAN_REQUIRE(x && frames, "stft_frames: NULL pointer");
  SEGAN_REQUIRE(B > 0 && T > 1 && hop > 0 && win >= 1 && win <= n_fft, "stft_frames: bad sizes");
  SEGAN_REQUIRE(n_fft / 2 < T, "stft_frames: reflect padding needs n_fft/2 < T");
  const int NF = 1 + T / hop;
  const int off = (n_fft - win) / 2 - n_fft / 2;
  const size_t total = (size_t)B * NF * win;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(stft_frames_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, frames,
                     B, T, NF, hop, win, off);
  return segan_check_launch("stft_frames");
}

// db[r][k] = 10*log10(re^2 + im^2 + eps),  S rows are [re(0..nbins) | im(0..nbins)]
__global__ void powdb_kernel(const float* __restrict__ S, float* __restrict__ db, size_t rows,
                             int nbins, int pitch, float eps) {
  const size_t total = rows * nbins;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / nbins;
    const int k = (int)(i - r * nbins);
    const float re = S[r * pitch + k], im = S[r * pitch + nbins + k];
    db[i] = 10.0f * log10f(fmaf(re, re, im * im) + eps);
  }
}

extern "C" int segan_powdb(const float* S, float* db, int64_t rows, int nbins, int pitch, float eps,
                           void* stream) {
  SEGAN_REQUIRE(pitch >= 2 * nbins, "powdb: pitch < 2*nbins");
  SEGAN_REQUIRE(S && db, "powdb: NULL pointer");
  SEGAN_REQUIRE(rows > 0 && nbins > 0, "powdb: bad sizes");
  const size_t total = (size_t)rows * nbins;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(powdb_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, db,
                     (size_t)rows, nbins, pitch, eps);
  return segan_check_launch("powdb");
}

// dS = ddb * d(10 log10(p + eps))/d(re, im) = ddb * (20/ln 10) * (re, im) / (p + eps)
__global__ void powdb_bwd_kernel(const float* __restrict__ S, const float* __restrict__ ddb,
                                 float* __restrict__ dS, size_t rows, int nbins, int pitch,
                                 float eps) {
  const size_t total = rows * nbins;
  const float c = 8.685889638065035f;   // 20 / ln(10)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / nbins;
    const int k = (int)(i - r * nbins);
    const float re = S[r * pitch + k], im = S[r * pitch + nbins + k];
    const float g = ddb[i] * c / (fmaf(re, re, im * im) + eps);
    dS[r * pitch + k] = g * re;
    dS[r * pitch + nbins + k] = g * im;
    if (k < pitch - 2 * nbins) dS[r * pitch + 2 * nbins + k] = 0.0f;   // pad columns
  }
}

extern "C" int segan_powdb_bwd(const float* S, const float* ddb, float* dS, int64_t rows, int nbins,
                               int pitch, float eps, void* stream) {
  SEGAN_REQUIRE(pitch >= 2 * nbins, "powdb_bwd: pitch < 2*nbins");
  SEGAN_REQUIRE(S && ddb && dS, "powdb_bwd: NULL pointer");
  SEGAN_REQUIRE(rows > 0 && nbins > 0, "powdb_bwd: bad sizes");
  const size_t total = (size_t)rows * nbins;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(powdb_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, ddb, dS,
                     (size_t)rows, nbins, pitch, eps);
  return segan_check_launch("powdb_bwd");
}

// dx[b][p] = sum of dframes[b*NF+f][j] over every (f, j) whose (reflected) source sample is p.
// Gather form: p is reached directly (o = p) and, near the edges, through the two mirrors
// (o = -p, o = 2(T-1) - p); for each of the three the frames f with 0 <= o - f*hop - off < win.
__global__ void stft_overlap_add_kernel(const float* __restrict__ dframes, float* __restrict__ dx,
                                        int B, int T, int NF, int hop, int win, int off) {
  const size_t total = (size_t)B * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % T);
    const int b = (int)(i / T);
    const float* df = dframes + (size_t)b * NF * win;
    float acc = 0.0f;
#pragma unroll
    for (int mirror = 0; mirror < 3; ++mirror) {
      int o;
      if (mirror == 0) o = p;
      else if (mirror == 1) { if (p == 0) continue; o = -p; }
      else { if (p == T - 1) continue; o = 2 * (T - 1) - p; }
      // f*hop + j + off = o, 0 <= j < win
      const int c = o - off;
      if (c < 0) continue;
      int f_hi = c / hop;
      if (f_hi > NF - 1) f_hi = NF - 1;
      int f_lo = c - win + 1;
      f_lo = f_lo <= 0 ? 0 : (f_lo + hop - 1) / hop;
      for (int f = f_lo; f <= f_hi; ++f) acc += df[(size_t)f * win + (c - f * hop)];
    }
    dx[i] = acc;
  }
}

extern "C" int segan_stft_overlap_add(const float* dframes, float* dx, int B, int T, int n_fft,
                                      int hop, int win, void* stream) {
  SEGAN_REQUIRE(dframes && dx, "stft_overlap_add: NULL pointer");
  SEGAN_REQUIRE(B > 0 && T > 1 && hop > 0 && win >= 1 && win <= n_fft, "stft_overlap_add: bad sizes");
  SEGAN_REQUIRE(n_fft / 2 < T, "stft_overlap_add: reflect padding needs n_fft/2 < T");
  const int NF = 1 + T / hop;
  const int off = (n_fft - win) / 2 - n_fft / 2;
  const size_t total = (size_t)B * T;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(stft_overlap_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     dframes, dx, B, T, NF, hop, win, off);
  return segan_check_launch("stft_overlap_add");
}

// ====================================================================================
// Input side: int16 PCM slices -> [-1,1] -> pre-emphasis, on the GPU.
// se_dataset.py:108-117,196-197: x = (2/65535)(pcm - 32767) + 1 ; y[n] = x[n] - coef*x[n-1]
// evaluated in float64 by numpy on the WHOLE wav before slicing, so a slice needs one sample
// of left context: every stored row holds T+1 samples (row[0] = the sample before the slice)
// and `first[b]` marks slices that start at sample 0 of their wav (y[0] = x[0] there).
// The arithmetic is done in double and rounded once, like the reference: bit-exact.
// ====================================================================================
__global__ void pcm16_prep_kernel(const int16_t* __restrict__ pcm, const unsigned char* __restrict__ first,
                                  float* __restrict__ clean, float* __restrict__ noisy, int B, int T,
                                  double coef) {
  const size_t total = (size_t)B * 2 * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % T);
    const size_t r = i / T;            // row = 2*b + which
    const int b = (int)(r >> 1);
    const int16_t* src = pcm + r * (size_t)(T + 1);
    const double x = (2.0 / 65535.0) * ((double)src[n + 1] - 32767.0) + 1.0;
    double y = x;
    if (coef > 0.0 && !(n == 0 && first[b])) {
      const double xp = (2.0 / 65535.0) * ((double)src[n] - 32767.0) + 1.0;
      y = x - coef * xp;
    }
    float* dst = (r & 1) ? noisy : clean;
    dst[(size_t)b * T + n] = (float)y;
  }
}

extern "C" int segan_pcm16_prep(const int16_t* pcm, const unsigned char* first, float* clean,
                                float* noisy, int B, int T, double coef, void* stream) {
  SEGAN_REQUIRE(pcm && first && clean && noisy, "pcm16_prep: NULL pointer");
  SEGAN_REQUIRE(B > 0 && T > 0, "pcm16_prep: bad sizes");
  const size_t total = (size_t)B * 2 * T;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  const double c = coef > 0.0 ? coef : 0.0;   // a double, like the reference's python float
  hipLaunchKernelGGL(pcm16_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pcm, first,
                     clean, noisy, B, T, c);
  return segan_check_launch("pcm16_prep");
}
