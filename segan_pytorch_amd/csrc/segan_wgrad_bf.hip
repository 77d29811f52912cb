// segan_wgrad_bf.hip — the W contraction form (both weight gradients) on the bf16 matrix
// cores, the companion of segan_conv_bf.hip:
//
//   dW[m, n, S*u + r] += sum_{b,t} lo[b, m, t] * HI_r[b, n, t + u]
//
// The fp32 kernel contracts over consecutive time positions; a 16-wide bf16 MFMA would then
// need its 8-element B fragment at the unaligned position t + u.  Here the contraction index
// of one MFMA is (time half, SAMPLE): a lane's 8 contiguous bf16 are the same (row, time)
// of 8 consecutive samples, so the fragment of tap u is simply the 16-byte piece at position
// t + u — aligned for every tap.  LDS keeps
//     lo tile  [plane][time q][row m]      16-B pieces (8 samples), rows XOR-swizzled by q
//     hi tile  [plane][virtual ch][pos]    16-B pieces, row pitch chosen per stride so the
//                                          (channel, tap) read pattern is conflict-free
// NPL = 1 ("bf16") or 3 ("bf16x3": exact 3-way split, six partial products) as in
// segan_conv_bf.hip.  A chunk is 8 samples x TQ time positions; the contraction is split
// over blockIdx.z and reduced with fp32 atomics like the fp32 kernel.
#include "segan_conv_shared.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wsplit3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

// hi-tile row pitch (pieces): >= PW and = 8 (mod 16) for U = 8, = 0 (mod 16) for U = 16
__host__ __device__ constexpr int wbf_pitch(int U, int PW) {
  return U == 8 ? ((PW + 7) / 16) * 16 + 8 : (U == 16 ? ((PW + 15) / 16) * 16 : PW + (PW & 1));
}

// LO_MODE: 0 = lo converted in the kernel with its transform, 1 = same, identity transform,
// 2 = lo pre-packed by wgrad_pack_lo_kernel (bf16 planes, [sample group][time][row][8]).
// HI_ID: hi has the identity transform (the gradient operand always has).
template <int U, int NPL, int TQ, int LO_MODE, bool HI_ID>
__global__ __launch_bounds__(256, 2) void wgrad_bf_kernel(const WgradArgs a) {
  constexpr int S = 32 / U;
  constexpr int MB = 128;
  constexpr int CVW = 128 / U;         // virtual channels per block
  constexpr int PW = TQ + U - 1;       // hi positions per chunk
  constexpr int QW = wbf_pitch(U, PW);
  constexpr int F4 = TQ / 4;           // float4 loads per lo row and chunk
  constexpr int NAU = (MB * F4) / 256; // lo units per thread
  constexpr int NBU = (CVW * PW + 255) / 256;
  constexpr int SWM = LO_MODE == 2 ? 0 : 8 / F4;   // swizzle step (packed lo is written row-linear)
  constexpr bool LO_ID = LO_MODE == 1;
  constexpr bool LO_PK = LO_MODE == 2;
  constexpr int NPP = (TQ * MB) / 256;  // packed lo: 16-B pieces per thread and plane
  static_assert(QW >= PW, "pitch");
  static_assert(NAU >= 1, "TQ >= 8");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* Al = reinterpret_cast<u32x4*>(smem_raw);     // [NPL][TQ][MB]
  u32x4* Bl = Al + NPL * TQ * MB;                      // [NPL][CVW][QW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int cv0 = blockIdx.x * CVW;
  const int m0 = blockIdx.y * MB;
  const int nchunks = ((a.B + 7) / 8) * a.bf_qc;
  const int c_beg = blockIdx.z * a.bf_cps;
  const int c_end = min(c_beg + a.bf_cps, nchunks);
  if (c_beg >= c_end) return;
  const int Ls = a.Ls;

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

  // ---- lo staging units: (row m, float4 f) -> 8 samples x 4 time positions ----
  // addresses are segment-0 base (wave-uniform, SGPRs) + a 32-bit byte offset; rows of
  // segment 1 carry the byte distance to it in a_adj (arithmetic select, no pointer array)
  unsigned a_ro[NAU];       // row offset (elements) inside its segment
  unsigned a_cs[NAU];       // sample stride of that segment
  long a_adj[NAU];
  bool a_rok[NAU];
  ChanXf a_xf[NAU];
  int a_f[NAU], a_m[NAU];
  const long lo_delta = a.lo.C1 > 0 ? reinterpret_cast<const char*>(a.lo.p1) -
                                          reinterpret_cast<const char*>(a.lo.p0) : 0L;
  const long hi_delta = a.hi.C1 > 0 ? reinterpret_cast<const char*>(a.hi.p1) -
                                          reinterpret_cast<const char*>(a.hi.p0) : 0L;
#pragma unroll
  for (int k = 0; k < NAU; ++k) {
    const int id = tid + 256 * k;
    a_f[k] = id % F4;
    a_m[k] = id / F4;
    int m = m0 + a_m[k];
    a_rok[k] = m < a.M;
    m = a_rok[k] ? m : 0;
    const bool s1 = m >= a.lo.C0;
    a_ro[k] = (unsigned)(s1 ? m - a.lo.C0 : m) * (unsigned)Ls;
    a_cs[k] = (unsigned)(s1 ? a.lo.C1 : a.lo.C0) * (unsigned)Ls;
    a_adj[k] = s1 ? lo_delta : 0L;
    a_xf[k] = segan_chan_xf(a.lo, m);
  }
  // ---- hi staging units: (virtual channel, position) -> 8 samples ----
  unsigned b_ro[NBU], b_cs[NBU];
  long b_adj[NBU];
  int b_pos[NBU], b_r[NBU], b_cvl[NBU];
  bool b_cok[NBU];
  ChanXf b_xf[NBU];
#pragma unroll
  for (int k = 0; k < NBU; ++k) {
    const int id = tid + 256 * k;
    const int nl = id / (S * PW);
    const int x = id - nl * (S * PW);
    b_pos[k] = x / S;
    b_r[k] = x % S;
    b_cvl[k] = nl * S + b_r[k];
    const bool uok = id < CVW * PW;
    int n = cv0 / S + nl;
    b_cok[k] = uok && n < a.N;
    n = b_cok[k] ? n : 0;
    const bool s1 = n >= a.hi.C0;
    b_ro[k] = (unsigned)(s1 ? n - a.hi.C0 : n) * (unsigned)a.Lhi;
    b_cs[k] = (unsigned)(s1 ? a.hi.C1 : a.hi.C0) * (unsigned)a.Lhi;
    b_adj[k] = s1 ? hi_delta : 0L;
    b_xf[k] = segan_chan_xf(a.hi, n);
    if (!uok) { b_pos[k] = 0; b_cvl[k] = 0; }
  }

  f32x4 areg[LO_PK ? 1 : NAU][LO_PK ? 1 : 8];
  u32x4 apk[LO_PK ? NPL : 1][LO_PK ? NPP : 1];
  unsigned apk_ok = 0u;
  float breg[NBU][8];
  bool a_qok[NAU];
  bool b_iok[NBU];
  int cur_b0 = 0;

  // A chunk whose 8 samples all exist (every chunk but those of the last sample group) takes
  // the fast path: no per-sample clamps or masks.  `full` is wave-uniform.
  bool cur_full = true;
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    const int sg = c / a.bf_qc;
    const int q0 = (c - sg * a.bf_qc) * TQ;
    const int b0 = 8 * sg;
    cur_b0 = b0;
    cur_full = b0 + 8 <= a.B;
    if (LO_PK) {
      // piece id = tid + 256*i -> (time q = id / MB, row m = id % MB): 64 consecutive rows per
      // wave instruction = 1 KB contiguous
      const char* base = reinterpret_cast<const char*>(a.lo_pk) +
                         ((size_t)sg * Ls * a.Mp + m0) * 16;
      apk_ok = 0u;
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        const int id = tid + 256 * i;
        const int q = q0 + id / MB;
        const bool ok = q < Ls;
        if (ok) apk_ok |= 1u << i;
        const size_t off = ((size_t)(ok ? q : 0) * a.Mp + id % MB) * 16;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          apk[LO_PK ? p : 0][LO_PK ? i : 0] =
              *reinterpret_cast<const u32x4*>(base + p * a.lo_pk_plane + off);
      }
    }
#pragma unroll
    for (int k = 0; k < (LO_PK ? 0 : NAU); ++k) {
      const int q = q0 + 4 * a_f[k];
      a_qok[k] = q < Ls;
      const unsigned o0 = a_ro[k] + (a_qok[k] ? q : 0) + (unsigned)b0 * a_cs[k];
      const char* bp = reinterpret_cast<const char*>(a.lo.p0) + a_adj[k] + (size_t)(unsigned)(o0 << 2);
      const size_t st = (size_t)a_cs[k] << 2;
      if (cur_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) areg[k][e] = *reinterpret_cast<const f32x4*>(bp + e * st);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          areg[k][e] = *reinterpret_cast<const f32x4*>(bp + ((b0 + e < a.B) ? e * st : 0));
      }
    }
#pragma unroll
    for (int k = 0; k < NBU; ++k) {
      const int idx = segan_hi_index(S * (q0 + b_pos[k]) + b_r[k], a.Lhi, a.padL, a.mode, a.roll);
      b_iok[k] = idx >= 0;
      const unsigned o0 = b_ro[k] + (b_iok[k] ? idx : 0) + (unsigned)b0 * b_cs[k];
      const char* bp = reinterpret_cast<const char*>(a.hi.p0) + b_adj[k] + (size_t)(unsigned)(o0 << 2);
      const size_t st = (size_t)b_cs[k] << 2;
      if (cur_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) breg[k][e] = *reinterpret_cast<const float*>(bp + e * st);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          breg[k][e] = *reinterpret_cast<const float*>(bp + ((b0 + e < a.B) ? e * st : 0));
      }
    }
  };
  // 8 values -> NPL bf16 planes, ANDed with a lane mask (0 / ~0) so invalid rows / positions
  // become zeros at 4 dword ops per piece instead of one select per element
  auto to_planes = [&](const float (&v)[8], unsigned lm, u32x4 (&out)[3]) __attribute__((always_inline)) {
    bf16x8 pl[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 p1, p2, p3;
      wsplit3(v[e], p1, p2, p3);
      pl[0][e] = p1; pl[1][e] = p2; pl[2][e] = p3;
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      out[p] = __builtin_bit_cast(u32x4, pl[p]);
#pragma unroll
      for (int d = 0; d < 4; ++d) out[p][d] &= lm;
    }
  };
  auto store_chunk = [&]() __attribute__((always_inline)) {
    const int nvalid = a.B - cur_b0;   // samples e < nvalid exist
    if (LO_PK) {
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        const int id = tid + 256 * i;
        const unsigned lm = ((apk_ok >> i) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          u32x4 v = apk[LO_PK ? p : 0][LO_PK ? i : 0];
#pragma unroll
          for (int d = 0; d < 4; ++d) v[d] &= lm;
          Al[(p * TQ + id / MB) * MB + id % MB] = v;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < (LO_PK ? 0 : NAU); ++k) {
      const unsigned lm = (a_rok[k] && a_qok[k]) ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = areg[k][e][i];
          if (!LO_ID) {
            v[e] = fmaf(v[e], a_xf[k].sc, a_xf[k].sh);
            v[e] = fmaf(a_xf[k].sl, fminf(v[e], 0.0f), fmaxf(v[e], 0.0f));
          }
        }
        if (!cur_full) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = e < nvalid ? v[e] : 0.0f;
        }
        u32x4 out[3];
        to_planes(v, lm, out);
        const int q = 4 * a_f[k] + i;
        const int mm = a_m[k] ^ (a_f[k] * SWM);
#pragma unroll
        for (int p = 0; p < NPL; ++p) Al[(p * TQ + q) * MB + mm] = out[p];
      }
    }
#pragma unroll
    for (int k = 0; k < NBU; ++k) {
      if (tid + 256 * k >= CVW * PW) continue;
      const unsigned lm = (b_cok[k] && b_iok[k]) ? 0xFFFFFFFFu : 0u;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = breg[k][e];
        if (!HI_ID) {
          v[e] = fmaf(v[e], b_xf[k].sc, b_xf[k].sh);
          v[e] = fmaf(b_xf[k].sl, fminf(v[e], 0.0f), fmaxf(v[e], 0.0f));
        }
      }
      if (!cur_full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < nvalid ? v[e] : 0.0f;
      }
      u32x4 out[3];
      to_planes(v, lm, out);
#pragma unroll
      for (int p = 0; p < NPL; ++p) Bl[(p * CVW + b_cvl[k]) * QW + b_pos[k]] = out[p];
    }
  };

  load_chunk(c_beg);
  store_chunk();
  __syncthreads();
  for (int c = c_beg; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) load_chunk(c + 1);
#pragma unroll
    for (int kk = 0; kk < TQ / 2; ++kk) {
      // lane half h contracts time position q = kk + h*TQ/2; its rows sit at m ^ swz(q)
      const int qa = kk + h * (TQ / 2);
      const int sw = LO_PK ? 0 : ((qa >> 2) % F4) * SWM;
      bf16x8 af[2][NPL], bf[2][NPL];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          af[i][p] = __builtin_bit_cast(bf16x8, Al[(p * TQ + qa) * MB + (arow[i] ^ sw)]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          bf[j][p] = __builtin_bit_cast(bf16x8, Bl[p * CVW * QW + bbase[j] + kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NPL == 1) {
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
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
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


// ====================================================================================
// lo pre-pack: fp32 lo[b][m][t] (with its transform) -> bf16 planes [p][sg][t][Mp][8 samples].
// Every column tile of the weight gradient re-reads the lo operand (N*S*U/128 = 32..256
// times), so it is converted ONCE here: the hot kernel then streams 16-byte pieces in full
// lines (2 B/element instead of 4) and spends no VALU on it.  Tile 64 time steps x 16 rows
// through LDS so that both the reads (256 B per row run) and the writes (256 B per time step)
// are coalesced.
// ====================================================================================
template <int NPL>
__global__ __launch_bounds__(256) void wgrad_pack_lo_kernel(const segan_src lo, __bf16* __restrict__ out,
                                                            size_t plane_elems, int B, int M, int Mp,
                                                            int Ls, int identity) {
  __shared__ u32x4 tile[NPL][16][65];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * 64, m0 = blockIdx.y * 16, sg = blockIdx.z;
  {
    const int ql = tid & 63;
    const int q = q0 + ql;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int ml = (tid >> 6) + 4 * pass;
      const int m = m0 + ml;
      const bool ok = m < M && q < Ls;
      const int mc = ok ? m : 0;
      const bool s1 = mc >= lo.C0;
      const float* row = s1 ? lo.p1 + (size_t)(mc - lo.C0) * Ls : lo.p0 + (size_t)mc * Ls;
      const size_t cs = (size_t)(s1 ? lo.C1 : lo.C0) * Ls;
      const ChanXf xf = segan_chan_xf(lo, mc);
      bf16x8 pl[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int b = 8 * sg + e;
        float v = row[(size_t)(b < B ? b : 0) * cs + (ok ? q : 0)];
        if (!identity) {
          v = fmaf(v, xf.sc, xf.sh);
          v = fmaf(xf.sl, fminf(v, 0.0f), fmaxf(v, 0.0f));
        }
        v = (ok && b < B) ? v : 0.0f;
        __bf16 p1, p2, p3;
        wsplit3(v, p1, p2, p3);
        pl[0][e] = p1; pl[1][e] = p2; pl[2][e] = p3;
      }
#pragma unroll
      for (int p = 0; p < NPL; ++p) tile[p][ml][ql] = __builtin_bit_cast(u32x4, pl[p]);
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

template <int U, int NPL, int LO_MODE, bool HI_ID>
static int launch_wgrad_bf_x(WgradArgs& a, hipStream_t st) {
  constexpr int TQ = NPL == 3 ? 8 : 16;
  constexpr int CVW = 128 / U;
  constexpr int QW = wbf_pitch(U, TQ + U - 1);
  if (a.Ls % 4 != 0 || 2 * a.Ls < TQ) {
    segan_set_error("wgrad_bf: low-rate length %d stays on the fp32 kernel", a.Ls);
    return SEGAN_EUNSUPPORTED;
  }
  if ((long)a.B * a.M * a.Ls >= (1L << 30) || (long)a.B * a.N * a.Lhi >= (1L << 30)) {
    segan_set_error("wgrad_bf: operand exceeds the 2^30 element indexing limit");
    return SEGAN_EUNSUPPORTED;
  }
  const bool lo_identity = !a.lo.scale && !a.lo.shift && !a.lo.slope;
  if (int e = segan_src_defaults(&a.lo, st, "wgrad(lo)")) return e;
  if (int e = segan_src_defaults(&a.hi, st, "wgrad(hi)")) return e;
  if (LO_MODE == 2) {
    a.Mp = round_up(a.M, 128);
    const int SG = ceil_div(a.B, 8);
    const size_t plane_elems = (size_t)SG * a.Ls * a.Mp * 8;
    a.lo_pk_plane = plane_elems * sizeof(__bf16);
    hipLaunchKernelGGL((wgrad_pack_lo_kernel<NPL>), dim3(ceil_div(a.Ls, 64), a.Mp / 16, SG), dim3(256),
                       0, st, a.lo, reinterpret_cast<__bf16*>(a.lo_pk), plane_elems, a.B, a.M, a.Mp,
                       a.Ls, lo_identity ? 1 : 0);
    if (int e = segan_check_launch("wgrad_pack_lo_kernel")) return e;
  }
  a.bf_qc = ceil_div(a.Ls, TQ);
  const int chunks = ceil_div(a.B, 8) * a.bf_qc;
  const int ncol = ceil_div(a.Cv, CVW);
  const int nrow = ceil_div(a.M, 128);
  const int tiles = ncol * nrow;
  int nsplit = ceil_div(1536, tiles);
  if (nsplit > chunks / 4) nsplit = chunks / 4;
  if (nsplit < 1) nsplit = 1;
  a.bf_cps = ceil_div(chunks, nsplit);
  nsplit = ceil_div(chunks, a.bf_cps);
  const size_t lds = (size_t)(NPL * TQ * 128 + NPL * CVW * QW) * 16;
  auto kern = wgrad_bf_kernel<U, NPL, TQ, LO_MODE, HI_ID>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(ncol, nrow, nsplit), dim3(256), lds, st, a);
  return segan_check_launch("wgrad_bf_kernel");
}

template <int U, int NPL>
static int launch_wgrad_bf(WgradArgs& a, hipStream_t st) {
  const bool lo_id = !a.lo.scale && !a.lo.shift && !a.lo.slope;
  const bool hi_id = !a.hi.scale && !a.hi.shift && !a.hi.slope;
  if (a.lo_pk != nullptr)
    return hi_id ? launch_wgrad_bf_x<U, NPL, 2, true>(a, st) : launch_wgrad_bf_x<U, NPL, 2, false>(a, st);
  if (lo_id && hi_id) return launch_wgrad_bf_x<U, NPL, 1, true>(a, st);
  if (lo_id) return launch_wgrad_bf_x<U, NPL, 1, false>(a, st);
  if (hi_id) return launch_wgrad_bf_x<U, NPL, 0, true>(a, st);
  return launch_wgrad_bf_x<U, NPL, 0, false>(a, st);
}

size_t segan_wgrad_bf_scratch_bytes(int B, int M, int Ls, int planes) {
  return (size_t)planes * ceil_div(B, 8) * Ls * round_up(M, 128) * 8 * sizeof(__bf16);
}

int segan_wgrad_bf(WgradArgs& a, int U, int planes, hipStream_t st) {
  if (U == 8) return planes == 3 ? launch_wgrad_bf<8, 3>(a, st) : launch_wgrad_bf<8, 1>(a, st);
  if (U == 16) return planes == 3 ? launch_wgrad_bf<16, 3>(a, st) : launch_wgrad_bf<16, 1>(a, st);
  return planes == 3 ? launch_wgrad_bf<32, 3>(a, st) : launch_wgrad_bf<32, 1>(a, st);
}
