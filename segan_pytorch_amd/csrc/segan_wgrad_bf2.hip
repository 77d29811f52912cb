// segan_wgrad_bf2.hip — round-3 form of the W contraction (both weight gradients) on the bf16
// matrix cores, the companion of segan_conv_bf2.hip:
//
//   dW[m, n, S*u + r] += sum_{b,t} lo[b, m, t] * HI_r[b, n, t + u]
//
// As in round 1 the contraction index of one MFMA is (time half, 8 SAMPLES): a lane's 8 contiguous
// bf16 are the same (row, time) of 8 consecutive samples, so tap u of the hi operand is the
// 16-byte piece at position t + u — aligned for every tap.  What is new: BOTH operands are packed
// once per call into that piece order
//
//   lo  Lp[plane][sample group][time t][row m (pitch Mp)][8 samples]        (wgrad_pack_lo2_kernel)
//   hi  Hp[plane][sample group][virtual channel (n, r)][position q][8]      (wgrad_pack_hi_kernel:
//       q in [0, Ls + U - 1), value pad(roll(hi))[n, S*q + r] with the transform applied)
//
// and reach LDS by LDS-DMA: per chunk of TQ time steps 2*TQ instructions for the 128-row lo tile,
// 6 for the 16-virtual-channel hi window (lane-linear in LDS = [channel][pitch 24]); the loop
// then holds nothing but ds_read_b128 and MFMAs.  Round 1's kernel converted the hi window from
// fp32 inside the loop (8 VALU per MFMA: 0.2 of the bf16 peak).
#include "segan_conv_shared.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wsplit3b(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

// ---- lo: fp32 lo[b][m][t] (with its transform) -> planes [p][sg][t][Mp][8 samples] ----------
// 64 time steps x 16 rows through LDS: a thread reads 4 consecutive time steps of its row for the
// 8 samples as float4 (16 lanes = one 256-byte run per row and sample), the block writes 16 rows
// x 16 bytes = 256-byte runs per time step
template <int NPL>
__global__ __launch_bounds__(256) void wgrad_pack_lo2_kernel(const segan_src lo, __bf16* __restrict__ out,
                                                             size_t plane_elems, int B, int M, int Mp,
                                                             int Ls, int identity) {
  __shared__ u32x4 tile[NPL][16][65];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * 64, m0 = blockIdx.y * 16, sg = blockIdx.z;
  {
    const int ml = tid >> 4, tq = tid & 15;
    const int m = m0 + ml;
    const int q = q0 + 4 * tq;
    const bool rok = m < M;
    const int mc = rok ? m : 0;
    const bool s1 = mc >= lo.C0;
    const float* row = s1 ? lo.p1 + (size_t)(mc - lo.C0) * Ls : lo.p0 + (size_t)mc * Ls;
    const size_t cs = (size_t)(s1 ? lo.C1 : lo.C0) * Ls;
    const ChanXf xf = segan_chan_xf(lo, mc);
    const bool vec = (Ls & 3) == 0 && q + 3 < Ls;     // rows are 16-byte aligned when Ls % 4 == 0
    float v[8][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int b = 8 * sg + e;
      const float* src = row + (size_t)(b < B ? b : 0) * cs;
      if (vec) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + q);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e][i] = t[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e][i] = (q + i < Ls) ? src[q + i] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!identity) v[e][i] = segan_apply_xf(xf, v[e][i]);
        v[e][i] = (rok && b < B && q + i < Ls) ? v[e][i] : 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 pl[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 p1, p2, p3;
        wsplit3b(v[e][i], p1, p2, p3);
        pl[0][e] = p1; pl[1][e] = p2; pl[2][e] = p3;
      }
#pragma unroll
      for (int p = 0; p < NPL; ++p) tile[p][ml][4 * tq + i] = __builtin_bit_cast(u32x4, pl[p]);
    }
  }
  __syncthreads();
  {
    const int mw = tid & 15;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int qw = (tid >> 4) + 16 * pass;
      const int q = q0 + qw;
      if (q >= Ls) continue;
      const size_t piece = ((size_t)sg * Ls + q) * Mp + m0 + mw;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        *reinterpret_cast<u32x4*>(out + p * plane_elems + piece * 8) = tile[p][mw][qw];
    }
  }
}

// ---- hi: fp32 hi[b][n][Lhi] -> planes [p][sg][cv = n*S + r][q][8 samples] --------------------
// one thread = the S pieces (phases) of one (channel n, position q): the S padded samples S*q ..
// S*q + S-1 of a row are consecutive in memory except at the reflected ends and the roll's wrap
// point, so a lane reads them as ONE vector per sample (lanes along q: fully coalesced lines)
// and writes one piece per phase (each phase row contiguous along q).
template <int S, int NPL>
__global__ __launch_bounds__(256) void wgrad_pack_hi_kernel(const segan_src hi, __bf16* __restrict__ out,
                                                            size_t plane_elems, int B, int N, int Cvp,
                                                            int Lhi, int Tq, int padL, int mode,
                                                            int roll, int identity) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y, sg = blockIdx.z;
  if (q >= Tq) return;
  int idx[S];
#pragma unroll
  for (int r = 0; r < S; ++r) idx[r] = segan_hi_index(S * q + r, Lhi, padL, mode, roll);
  bool run = n < N && idx[0] >= 0;          // S consecutive stored samples?
#pragma unroll
  for (int r = 1; r < S; ++r) run = run && idx[r] == idx[0] + r;
  const int nc = n < N ? n : 0;
  const bool s1 = nc >= hi.C0;
  const float* row = s1 ? hi.p1 + (size_t)(nc - hi.C0) * Lhi : hi.p0 + (size_t)nc * Lhi;
  const size_t cs = (size_t)(s1 ? hi.C1 : hi.C0) * Lhi;
  const ChanXf xf = segan_chan_xf(hi, nc);
  float v[8][S];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int b = 8 * sg + e;
    const float* src = row + (size_t)(b < B ? b : 0) * cs;
    if (run) {
      if (S == 4) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f4u t = *reinterpret_cast<const f4u*>(src + idx[0]);
        v[e][0] = t[0]; v[e][1] = t[1]; v[e][2 % S] = t[2]; v[e][3 % S] = t[3];
      } else if (S == 2) {
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        const f2u t = *reinterpret_cast<const f2u*>(src + idx[0]);
        v[e][0] = t[0]; v[e][1 % S] = t[1];
      } else {
        v[e][0] = src[idx[0]];
      }
    } else {
#pragma unroll
      for (int r = 0; r < S; ++r) v[e][r] = (n < N && idx[r] >= 0) ? src[idx[r]] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < S; ++r) {
      if (!identity) v[e][r] = segan_apply_xf(xf, v[e][r]);
      const bool ok = n < N && b < B && (run || idx[r] >= 0);
      v[e][r] = ok ? v[e][r] : 0.0f;
    }
  }
#pragma unroll
  for (int r = 0; r < S; ++r) {
    bf16x8 pl[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 p1, p2, p3;
      wsplit3b(v[e][r], p1, p2, p3);
      pl[0][e] = p1; pl[1][e] = p2; pl[2][e] = p3;
    }
    const int cv = n * S + r;
    if (cv >= Cvp) continue;
    const size_t piece = ((size_t)sg * Cvp + cv) * Tq + q;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
      *reinterpret_cast<u32x4*>(out + p * plane_elems + piece * 8) = __builtin_bit_cast(u32x4, pl[p]);
  }
}

// hi-tile row pitch (pieces): >= TQ + U - 1 and = 8 (mod 16) for U = 8, = 0 (mod 16) for U = 16
__host__ __device__ constexpr int wbf2_pitch(int U, int PW) {
  return U == 8 ? ((PW + 7) / 16) * 16 + 8 : (U == 16 ? ((PW + 15) / 16) * 16 : ((PW + 15) / 16) * 16);
}

struct Wbf2Args {
  const __bf16* lo;       // packed lo planes
  const __bf16* hi;       // packed hi planes
  long lo_plane_bytes, hi_plane_bytes;
  float* dw;
  int M, N, K, Cv, Cvp, Mp, Ls, Tq;
  int qc;                 // time chunks per sample group
  int cps;                // chunks per workgroup (split of the contraction)
  int nchunks;
};

template <int U, int NPL, int TQ>
struct WGeom {
  static constexpr int S = 32 / U;
  static constexpr int MB = 128;
  static constexpr int CVW = 128 / U;         // virtual channels per block
  static constexpr int PW = TQ + U - 1;       // hi positions per chunk
  static constexpr int QW = wbf2_pitch(U, PW);
  static constexpr int APIECES = NPL * TQ * MB;
  static constexpr int AINS = NPL * TQ * 2;                 // lo DMA instructions per chunk
  static constexpr int BINS = (CVW * QW + 63) / 64;         // hi DMA instructions per chunk and plane
  static constexpr int BK = (BINS + 3) / 4;                 // ... per wave
  static constexpr int BPAD = NPL * BINS * 64;
  static_assert(AINS % 4 == 0, "lo instructions divide among the 4 waves");
  static_assert(QW >= PW, "pitch");
};

template <int U, int NPL, int TQ>
__global__ __launch_bounds__(256, 2) void wgrad_bf2_kernel(const Wbf2Args a) {
  using G = WGeom<U, NPL, TQ>;
  constexpr int S = G::S, MB = G::MB, CVW = G::CVW, PW = G::PW, QW = G::QW, APIECES = G::APIECES;
  constexpr int BINS = G::BINS, BPAD = G::BPAD;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* Al0 = reinterpret_cast<u32x4*>(smem_raw);     // [2][NPL][TQ][MB]
  u32x4* Bl0 = Al0 + 2 * APIECES;                        // [2][NPL][CVW*QW rounded to 64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int cv0 = blockIdx.x * CVW;
  const int m0 = blockIdx.y * MB;
  const int c_beg = blockIdx.z * a.cps;
  const int c_end = min(c_beg + a.cps, a.nchunks);
  if (c_beg >= c_end) return;

  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(a.lo), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(a.hi), 0, 0x7fffffff, 0x00020000);

  // lo: instruction (p, q, i) moves rows 64*i + lane of time q; this wave takes i = wave & 1 and
  // the times q = (wave >> 1) + 2k
  const int li = wave & 1, lq = wave >> 1;
  const int lvo = (m0 + 64 * li + lane) * 16;
  // hi: LDS piece 64*k + lane = (channel, position) at pitch QW; this wave takes k = wave, wave + 4
  static_assert(G::BK <= 2, "two hi instructions per wave at most");
  auto hi_off = [&](int k) {
    const int pc = 64 * (wave + 4 * k) + lane;
    const int cvl = pc / QW, pos = pc - cvl * QW;
    return (cvl < CVW && pos < PW && cv0 + cvl < a.Cvp) ? ((cv0 + cvl) * a.Tq + pos) * 16
                                                        : (int)0x80000000u;
  };
  const int hvo0 = hi_off(0), hvo1 = hi_off(1);

  // ---- MFMA operand offsets (pieces) ----
  int arow[2], bbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) arow[i] = wm * 64 + 32 * i + l31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cc = wn * 64 + 32 * j + l31;
    bbase[j] = (cc / U) * QW + cc % U + h * (TQ / 2);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  auto dma_chunk = [&](int c, int buf) __attribute__((always_inline)) {
    const int sg = c / a.qc;
    const int q0 = (c - sg * a.qc) * TQ;
    u32x4* Al = Al0 + buf * APIECES;
    u32x4* Bl = Bl0 + buf * BPAD;
    const long lbase = ((long)sg * a.Ls + q0) * a.Mp * 16;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int k = 0; k < TQ / 2; ++k) {
        const int q = lq + 2 * k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            lrs, (__attribute__((address_space(3))) void*)(Al + (p * TQ + q) * MB + 64 * li), 16, lvo,
            (int)(p * a.lo_plane_bytes + lbase + (long)q * a.Mp * 16), 0, 0);
      }
    }
    const long hbase = ((long)sg * a.Cvp * a.Tq + q0) * 16;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int k = 0; k < G::BK; ++k) {
        if (wave + 4 * k < G::BINS)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              hrs, (__attribute__((address_space(3))) void*)(Bl + p * G::BINS * 64 + 64 * (wave + 4 * k)), 16,
              k == 0 ? hvo0 : hvo1, (int)(p * a.hi_plane_bytes + hbase), 0, 0);
      }
    }
  };

  dma_chunk(c_beg, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
  __syncthreads();
  for (int c = c_beg; c < c_end; ++c) {
    const int buf = (c - c_beg) & 1;
    if (c + 1 < c_end) dma_chunk(c + 1, buf ^ 1);
    const u32x4* Al = Al0 + buf * APIECES;
    const u32x4* Bl = Bl0 + buf * BPAD;
#pragma unroll
    for (int kk = 0; kk < TQ / 2; ++kk) {
      // lane half h contracts time position q = kk + h*TQ/2
      const int qa = kk + h * (TQ / 2);
      bf16x8 af[2][NPL], bf[2][NPL];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          af[i][p] = __builtin_bit_cast(bf16x8, Al[(p * TQ + qa) * MB + arow[i]]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          bf[j][p] = __builtin_bit_cast(bf16x8, Bl[p * BINS * 64 + bbase[j] + kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (NPL == 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
          }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
  }

  // ---- epilogue: dw[m][n][S*u + r] += acc ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cc = wn * 64 + 32 * j + l31;
    const int cv = cv0 + cc / U;
    const int u = cc % U;
    const int n = cv / S, r = cv % S;
    const int k = S * u + r;
    if (cv >= a.Cv || k >= a.K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < a.M) atomicAdd(a.dw + ((size_t)m * a.N + n) * a.K + k, acc[i][j][e]);
      }
  }
}

// ---- launch ----------------------------------------------------------------------------------
static inline size_t wbf2_lo_bytes(int B, int M, int Ls, int planes) {
  return (size_t)planes * ceil_div(B, 8) * Ls * round_up(M, 128) * 16;
}
static inline size_t wbf2_hi_bytes(int B, int Cv, int Ls, int U, int planes) {
  return (size_t)planes * ceil_div(B, 8) * round_up(Cv, 16) * (size_t)(Ls + U - 1) * 16;
}

size_t segan_wgrad_bf2_scratch_bytes(int B, int M, int N, int Ls, int S, int planes) {
  const int U = 32 / S;
  return ((wbf2_lo_bytes(B, M, Ls, planes) + 255) & ~(size_t)255) + wbf2_hi_bytes(B, N * S, Ls, U, planes);
}

template <int U, int NPL>
static int launch_wbf2(WgradArgs& w, void* scratch, size_t scratch_bytes, hipStream_t st) {
  constexpr int S = 32 / U;
  // bf16: 8 time steps per chunk = 45 KB of LDS, three workgroups per CU (16 steps at two per CU
  // measured the same); bf16x3: three planes per time step fill the LDS at 4
  constexpr int TQ = NPL == 3 ? 4 : 8;
  constexpr int CVW = 128 / U;
  constexpr int QW = wbf2_pitch(U, TQ + U - 1);
  constexpr int BINS = (CVW * QW + 63) / 64;
  if (w.Ls % TQ != 0) {
    segan_set_error("wgrad_bf2: low-rate length %d is not a multiple of %d", w.Ls, TQ);
    return SEGAN_EUNSUPPORTED;
  }
  const size_t lo_b = wbf2_lo_bytes(w.B, w.M, w.Ls, NPL), hi_b = wbf2_hi_bytes(w.B, w.Cv, w.Ls, U, NPL);
  const size_t lo_pad = (lo_b + 255) & ~(size_t)255;
  if (scratch == nullptr || scratch_bytes < lo_pad + hi_b || lo_b >= (size_t)0x7fff0000 ||
      hi_b >= (size_t)0x7fff0000) {
    segan_set_error("wgrad_bf2: scratch of %zu bytes needed (operands below 2 GiB)", lo_pad + hi_b);
    return SEGAN_EUNSUPPORTED;
  }
  const bool lo_id = !w.lo.scale && !w.lo.shift && !w.lo.slope;
  const bool hi_id = !w.hi.scale && !w.hi.shift && !w.hi.slope;
  if (int e = segan_src_defaults(&w.lo, st, "wgrad(lo)")) return e;
  if (int e = segan_src_defaults(&w.hi, st, "wgrad(hi)")) return e;
  const int SG = ceil_div(w.B, 8);
  Wbf2Args a;
  a.lo = (const __bf16*)scratch;
  a.hi = (const __bf16*)((char*)scratch + lo_pad);
  a.dw = w.dw;
  a.M = w.M; a.N = w.N; a.K = w.K; a.Cv = w.Cv; a.Cvp = round_up(w.Cv, 16);
  a.Mp = round_up(w.M, 128); a.Ls = w.Ls; a.Tq = w.Ls + U - 1;
  a.lo_plane_bytes = (long)(lo_b / NPL);
  a.hi_plane_bytes = (long)(hi_b / NPL);
  hipLaunchKernelGGL((wgrad_pack_lo2_kernel<NPL>), dim3(ceil_div(w.Ls, 64), a.Mp / 16, SG), dim3(256), 0,
                     st, w.lo, (__bf16*)scratch, (size_t)a.lo_plane_bytes / 2, w.B, w.M, a.Mp, w.Ls,
                     lo_id ? 1 : 0);
  if (int e = segan_check_launch("wgrad_pack_lo2_kernel")) return e;
  hipLaunchKernelGGL((wgrad_pack_hi_kernel<S, NPL>), dim3(ceil_div(a.Tq, 256), a.Cvp / S, SG), dim3(256), 0, st,
                     w.hi, (__bf16*)((char*)scratch + lo_pad), (size_t)a.hi_plane_bytes / 2, w.B, w.N,
                     a.Cvp, w.Lhi, a.Tq, w.padL, w.mode, w.roll, hi_id ? 1 : 0);
  if (int e = segan_check_launch("wgrad_pack_hi_kernel")) return e;
  a.qc = w.Ls / TQ;
  a.nchunks = SG * a.qc;
  const int ncol = ceil_div(w.Cv, CVW);
  const int nrow = ceil_div(w.M, 128);
  const int tiles = ncol * nrow;
  // one round of resident workgroups (256 CUs x 3): every further split is another 128 x 128
  // tile of fp32 atomics (measured: 768 workgroups 194 us per call, 1536 218 us)
  int nsplit = ceil_div(768, tiles);
  if (nsplit > a.nchunks / 4) nsplit = a.nchunks / 4;
  if (nsplit < 1) nsplit = 1;
  a.cps = ceil_div(a.nchunks, nsplit);
  nsplit = ceil_div(a.nchunks, a.cps);
  const size_t lds = (size_t)(2 * NPL * TQ * 128 + 2 * NPL * BINS * 64) * 16;
  static bool attr_done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!attr_done[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf2_kernel<U, NPL, TQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((wgrad_bf2_kernel<U, NPL, TQ>), dim3(ncol, nrow, nsplit), dim3(256), lds, st, a);
  segan_note_wgrad_launch(3, ncol * nrow, nsplit, a.cps);
  return segan_check_launch("wgrad_bf2_kernel");
}

int segan_wgrad_bf2(WgradArgs& a, int U, int planes, void* scratch, size_t scratch_bytes, hipStream_t st) {
  if (U == 8) return planes == 3 ? launch_wbf2<8, 3>(a, scratch, scratch_bytes, st)
                                 : launch_wbf2<8, 1>(a, scratch, scratch_bytes, st);
  if (U == 16) return planes == 3 ? launch_wbf2<16, 3>(a, scratch, scratch_bytes, st)
                                  : launch_wbf2<16, 1>(a, scratch, scratch_bytes, st);
  segan_set_error("wgrad_bf2: stride 1 stays on the fp32 kernel");
  return SEGAN_EUNSUPPORTED;
}
