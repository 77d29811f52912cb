// segan_conv_bf.hip — the F and T contraction forms on the bf16 matrix cores (gfx950
// v_mfma_f32_32x32x16_bf16), as an alternative to the exact-fp32 MFMA kernels of
// segan_conv.hip.  Two precisions share one kernel:
//
//   NPL = 1  "bf16"    operands rounded to bf16, fp32 accumulation (BASELINE config 5:
//                      "bf16 mixed precision on CDNA4 MFMA", tolerance re-stated in the tests)
//   NPL = 3  "bf16x3"  every fp32 operand is split exactly into three bf16 planes
//                      x = x1 + x2 + x3 (8+8+8 mantissa bits) and the product is formed from
//                      the six partial products of weight >= 2^-18:
//                      a1b1 + a1b2 + a2b1 + a1b3 + a3b1 + a2b2, each exact in fp32 and
//                      accumulated in fp32 — fp32-class accuracy at 6/16 of the fp32-MFMA cost.
//
// Same tiling, staging discipline, stream-K hybrid and epilogues as the fp32 kernel.  What
// changes is the contraction ordering: one MFMA contracts 16 (virtual) input channels of ONE
// tap; a lane holds 8 consecutive channels (one 16-byte LDS read).  So LDS keeps the
// activation tile position-major with channels innermost, [plane][half][position][8ch], the
// weight tile [plane][tap][half][row][8ch], and the packed weights in HBM are pre-split
// planes in exactly that order.
#include "segan_conv_shared.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x -> (p1, p2, p3) with p1 = bf16(x), p2 = bf16(x - p1), p3 = bf16(x - p1 - p2)
__device__ __forceinline__ void split3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

// base[elem_off] with a 32-bit byte offset (tensors < 2^30 elements, checked by the launcher):
// a wave-uniform base then stays in SGPRs and the lane offset is one VGPR
__device__ __forceinline__ float ld_f32(const float* base, unsigned elem_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) +
                                         (size_t)(unsigned)(elem_off << 2));
}

struct BfExtra {
  const __bf16* wp3;    // packed planes
  long plane_stride;    // elements between planes
  int ngroups;          // channel groups of 16
};

template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, int SHIFTMASK, int NPL, int TU>
__global__ __launch_bounds__(256, NPL == 1 ? 3 : 2) void corr_bf_kernel(const CorrArgs a, const BfExtra x) {
  constexpr int S = 32 / U;
  constexpr int SI = IN_HI ? S : 1;
  constexpr int WN = 4 / WM;
  constexpr int NI = MB / (32 * WM);
  constexpr int NJ = NB / (32 * WN);
  constexpr int NPT = MB / S;
  constexpr int TCH = U / TU;          // weight stages per channel group
  constexpr int MAXT = 2;              // staging tasks per thread (2*RLs <= 512)
  constexpr int NSH = SHIFTMASK ? 2 : 1;
  static_assert(MB == 128, "one weight piece per thread");
  static_assert(!OUT_HI || WM == 1, "T form: one wave holds all phases");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int RLs = a.RLs;
  // Wl: [2 buffers][NPL][TU][2][MB] pieces of 16 B ; Il: [NPL][2][RLs] pieces of 16 B
  u32x4* Wl0 = reinterpret_cast<u32x4*>(smem_raw);
  u32x4* Il = Wl0 + 2 * NPL * TU * 2 * MB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int nst = x.ngroups * TCH;     // weight stages per tile
  int tileA = blockIdx.x;
  long unit = (long)blockIdx.x * a.sk_units;
  const long unit_end = min(unit + (long)a.sk_units, a.sk_total);
  for (;;) {
  int tile, c0, c1;
  if (tileA < a.sk_nfull) {
    tile = tileA; c0 = 0; c1 = nst;
    tileA += gridDim.x;
  } else if (unit < unit_end) {
    const int t = (int)(unit / nst);
    c0 = (int)(unit - (long)t * nst);
    c1 = min(nst, c0 + (int)(unit_end - unit));
    unit += c1 - c0;
    tile = a.sk_nfull + t;
  } else {
    break;
  }
  const bool partial = (c0 != 0) || (c1 != nst);
  const int rowtile = a.rt0 + tile / a.ncoltiles;
  const int coltile = tile % a.ncoltiles;
  const int m0 = rowtile * MB;
  const int n0 = rowtile * NPT;
  if (!OUT_HI) {
    if (a.out0 == nullptr && m0 + MB <= a.OC0) continue;
    if (a.out1 == nullptr && m0 >= a.OC0) continue;
  }
  const ColTile ct = make_coltile(coltile * NB, a.Tcols, NB);

  // ---- activation staging tasks: task t = (half g, position j), 8 channels each ----
  // tk_base*[k][r]: element offset of (sample, phase-r position) inside segment 0 / 1, so an
  // element's address is base + channel*Lin (one 24-bit mad); tk_ok: per-phase validity bits
  unsigned tk_base0[MAXT][SI];
  int tk_d[MAXT];          // segment 1: sample offset minus segment 0's
  unsigned tk_ok[MAXT];
  int tk_g[MAXT], tk_j[MAXT];
  const bool dual = a.in.C1 > 0;
  const long pdelta = dual ? reinterpret_cast<const char*>(a.in.p1) - reinterpret_cast<const char*>(a.in.p0) : 0L;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int t = tid + 256 * k;
    const int g = t >= RLs ? 1 : 0;
    const int j = t - g * RLs;
    tk_g[k] = g;
    tk_j[k] = j;
    tk_ok[k] = 0u;
#pragma unroll
    for (int r = 0; r < SI; ++r) tk_base0[k][r] = 0u;
    tk_d[k] = 0;
    if (t < 2 * RLs) {
      int s, tau;
      lds_pos_decode(ct, j, a.Tcols, a.H, s, tau);
      const int b = ct.b0 + s;
      if (b < a.B) {
        const unsigned bo0 = (unsigned)(b * a.in.C0) * (unsigned)a.Lin;
        tk_d[k] = b * (a.in.C1 - a.in.C0) * a.Lin;
        const int wq = tau + a.win_start;
        if (IN_HI) {
#pragma unroll
          for (int r = 0; r < SI; ++r) {
            const int idx = segan_hi_index(S * wq + r, a.Lin, a.padL, a.mode, a.roll);
            if (idx >= 0) {
              tk_base0[k][r] = bo0 + idx;
              tk_ok[k] |= 1u << r;
            }
          }
        } else if (wq >= 0 && wq < a.Lin) {
          tk_base0[k][0] = bo0 + wq;
          tk_ok[k] = 1u;
        }
      }
    }
  }
  const int Nreal = IN_HI ? a.Cv / S : a.Cv;     // real input channels
  constexpr int EPC = IN_HI ? S : 1;             // tile elements per real channel
  constexpr int NRC = 8 / EPC;                   // real channels per task
  constexpr bool XF_EARLY = NRC <= 4;            // prefetch the transforms with the data

  // ---- per-lane operand offsets (in 16-B pieces) ----
  int arow[NI], boff[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) arow[i] = 32 * (wm * NI + i) + l31;
  int col_b[NJ], col_t[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cl = wn * (NB / WN) + 32 * j + l31;
    const int col = ct.col0 + cl;
    if (col < a.Ctot) {
      const int b = col / a.Tcols;
      col_b[j] = b;
      col_t[j] = col - b * a.Tcols;
      boff[j] = cl + (b - ct.b0) * a.H;
    } else {
      col_b[j] = -1;
      col_t[j] = 0;
      boff[j] = 0;
    }
  }

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // ---- staging registers ----
  u32x4 wreg[NPL][TU];
  float ireg[MAXT][8];
  ChanXf ixf[MAXT][XF_EARLY ? NRC : 1];
  // weight piece of this thread: half g = tid / 128, row = tid % 128.  Byte offset inside a
  // stage = ((tu*2 + wg)*RP + wgrow)*16; the stage base is wave-uniform.
  const int wg = tid >> 7, wr = tid & 127;
  const int wgrow = OUT_HI ? (wr / NPT) * a.NP + n0 + wr % NPT : m0 + wr;
  const unsigned w_thr = (unsigned)(wg * a.RP + wgrow) * 16u;
  const unsigned w_tap = (unsigned)a.RP * 32u;                      // bytes between taps
  const size_t w_stage = (size_t)a.RP * (size_t)(TU * 32);          // bytes between stages
  const size_t w_plane = (size_t)x.plane_stride * sizeof(__bf16);

  auto load_w = [&](int st) __attribute__((always_inline)) {
    const char* wst = reinterpret_cast<const char*>(x.wp3) + (size_t)st * w_stage;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int tu = 0; tu < TU; ++tu)
        wreg[p][tu] = *reinterpret_cast<const u32x4*>(wst + p * w_plane + (w_thr + tu * w_tap));
  };
  auto store_w = [&](int buf) __attribute__((always_inline)) {
    u32x4* Wl = Wl0 + buf * (NPL * TU * 2 * MB);
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) Wl[((p * TU + tu) * 2 + wg) * MB + wr] = wreg[p][tu];
  };
  // channel of element e of a task in group cg: nb + e/EPC with nb = (16cg + 8g)/EPC
  auto load_in = [&](int cg) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
      const int nb = (16 * cg + 8 * tk_g[k]) / EPC;
#pragma unroll
      for (int c = 0; c < NRC; ++c) {
        const int n = min(nb + c, Nreal - 1);
        if (XF_EARLY && !a.in_identity) ixf[k][XF_EARLY ? c : 0] = segan_chan_xf(a.in, n);
        if (!dual) {
          const unsigned ro = __umul24((unsigned)n, (unsigned)a.Lin);
#pragma unroll
          for (int r = 0; r < EPC; ++r) ireg[k][c * EPC + r] = ld_f32(a.in.p0, ro + tk_base0[k][IN_HI ? r : 0]);
        } else {
          // segment 1 as a byte adjustment of the segment-0 address (arithmetic, so that the
          // compiler keeps both bases in registers instead of indexing a stack array)
          const bool seg1 = n >= a.in.C0;
          const unsigned ro = __umul24((unsigned)(n - (seg1 ? a.in.C0 : 0)), (unsigned)a.Lin);
          const long adj = seg1 ? pdelta + 4L * tk_d[k] : 0L;
          const char* bp = reinterpret_cast<const char*>(a.in.p0) + adj;
#pragma unroll
          for (int r = 0; r < EPC; ++r)
            ireg[k][c * EPC + r] = *reinterpret_cast<const float*>(
                bp + (size_t)(unsigned)((ro + tk_base0[k][IN_HI ? r : 0]) << 2));
        }
      }
    }
  };
  auto store_in = [&](int cg) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
      if (tid + 256 * k >= 2 * RLs) continue;
      const int nb = (16 * cg + 8 * tk_g[k]) / EPC;
      const int nval = Nreal - nb;            // channels c < nval exist
      bf16x8 pl[3];
#pragma unroll
      for (int c = 0; c < NRC; ++c) {
        ChanXf xf;
        if (!a.in_identity) {
          if (XF_EARLY) xf = ixf[k][XF_EARLY ? c : 0];
          else xf = segan_chan_xf(a.in, min(nb + c, Nreal - 1));
        }
#pragma unroll
        for (int r = 0; r < EPC; ++r) {
          const int e = c * EPC + r;
          float v = ireg[k][e];
          if (!a.in_identity) {
            v = fmaf(v, xf.sc, xf.sh);
            v = fmaf(xf.sl, fminf(v, 0.0f), fmaxf(v, 0.0f));
          }
          const bool ok = (c < nval) && ((tk_ok[k] >> (IN_HI ? r : 0)) & 1u);
          v = ok ? v : 0.0f;
          __bf16 p1, p2, p3;
          split3(v, p1, p2, p3);
          pl[0][e] = p1; pl[1][e] = p2; pl[2][e] = p3;
        }
      }
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        Il[(p * 2 + tk_g[k]) * RLs + tk_j[k]] = __builtin_bit_cast(u32x4, pl[p]);
    }
  };

  // ---- main loop over weight stages; the activation tile is re-staged per channel group ----
  {
    const int cg0 = c0 / TCH;
    load_in(cg0);
    load_w(c0);
    store_in(cg0);
    store_w(0);
    __syncthreads();
  }
  bool in_flight = false;     // the activation tile of the next channel group is being loaded
  for (int st = c0; st < c1; ++st) {
    const int buf = (st - c0) & 1;
    const int cg = st / TCH, tc = st - cg * TCH;
    const bool more = st + 1 < c1;
    const bool new_group = more && ((st + 1) % TCH == 0);
    if (more) load_w(st + 1);
    // issue the next group's activation loads at the FIRST stage of this group (not the last):
    // TCH stages of MFMA work cover their latency
    if (!in_flight && (cg + 1) * TCH < c1) {
      load_in(cg + 1);
      in_flight = true;
    }
    const u32x4* Wl = Wl0 + buf * (NPL * TU * 2 * MB);
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      const int u = tc * TU + tu;
      bf16x8 af[NI][NPL], bf[NSH][NJ][NPL];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          af[i][p] = __builtin_bit_cast(bf16x8, Wl[((p * TU + tu) * 2 + h) * MB + arow[i]]);
#pragma unroll
      for (int sh = 0; sh < NSH; ++sh)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            bf[sh][j][p] = __builtin_bit_cast(bf16x8, Il[(p * 2 + h) * RLs + boff[j] + u + sh]);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        // T form: row block i of the tile is phase r = 32*i/NPT (WM == 1)
        const int sh = SHIFTMASK ? ((SHIFTMASK >> ((32 * i) / NPT)) & 1) : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (NPL == 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[sh][j][0], acc[i][j], 0, 0, 0);
          } else {
            // smallest partial products first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[sh][j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[sh][j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[sh][j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[sh][j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[sh][j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[sh][j][0], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    if (new_group) {
      // every wave is done reading the activation tile of this group before it is replaced
      __syncthreads();
      store_in(cg + 1);
      in_flight = false;
    }
    if (more) store_w(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue (same as the fp32 kernel) ----
  const bool add_bias = (c0 == 0);
  if (!OUT_HI) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + 32 * (wm * NI + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= a.Rvalid) continue;
        float* dst;
        int oc, och;
        if (row < a.OC0) { dst = a.out0; oc = a.OC0; och = row; }
        else { dst = a.out1; oc = a.OC1; och = row - a.OC0; }
        if (dst == nullptr) continue;
        const float bs = (a.bias && add_bias) ? a.bias[row] : 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (col_b[j] < 0) continue;
          float v = acc[i][j][e] + bs;
          float* o = dst + ((size_t)col_b[j] * oc + och) * (size_t)a.Lout + col_t[j];
          if (partial) { atomicAdd(o, v); continue; }
          if (a.act == SEGAN_ACT_TANH) v = tanhf(v);
          *o = v;
        }
      }
    }
  } else {
    constexpr bool QUAD = (S == 4 && WM == 1 && NI == 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (col_b[j] < 0) continue;
        const int q = col_t[j];
        if (QUAD) {
          const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (n >= a.Nout) continue;
          const float bs = (a.bias && add_bias) ? a.bias[n] : 0.0f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[r][j][e] + bs;
            if (!partial && a.act == SEGAN_ACT_TANH) v[r] = tanhf(v[r]);
          }
          const size_t rowoff = (size_t)col_b[j] * a.Nout + n;
          const int i0 = 4 * q - a.o_padL;
          if (!partial && a.o_roll == 0 && i0 >= 0 && i0 + 3 < a.Lout && (a.o_padL & 3) == 0) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(a.out0 + rowoff * (size_t)a.Lout + i0) = o;
            continue;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int P = 4 * q + r;
            int ii = P - a.o_padL;
            if (ii >= 0 && ii < a.Lout) {
              if (a.o_roll != 0) {
                ii -= a.o_roll;
                if (ii < 0) ii += a.Lout;
                if (ii >= a.Lout) ii -= a.Lout;
              }
              float* o = a.out0 + rowoff * (size_t)a.Lout + ii;
              if (partial) atomicAdd(o, v[r]); else *o = v[r];
            } else if (a.halo != nullptr) {
              const int hl = a.o_padL + a.o_padR;
              float* o = nullptr;
              if (ii < 0) o = a.halo + rowoff * hl + P;
              else if (ii - a.Lout < a.o_padR) o = a.halo + rowoff * hl + a.o_padL + (ii - a.Lout);
              if (o) { if (partial) atomicAdd(o, v[r]); else *o = v[r]; }
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int rloc = 32 * (wm * NI + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int r = rloc / NPT;
            const int n = n0 + rloc % NPT;
            if (n >= a.Nout) continue;
            float v = acc[i][j][e] + ((a.bias && add_bias) ? a.bias[n] : 0.0f);
            if (!partial && a.act == SEGAN_ACT_TANH) v = tanhf(v);
            const int P = S * q + r;
            int ii = P - a.o_padL;
            const size_t rowoff = (size_t)col_b[j] * a.Nout + n;
            if (ii >= 0 && ii < a.Lout) {
              if (a.o_roll != 0) {
                ii -= a.o_roll;
                if (ii < 0) ii += a.Lout;
                if (ii >= a.Lout) ii -= a.Lout;
              }
              float* o = a.out0 + rowoff * (size_t)a.Lout + ii;
              if (partial) atomicAdd(o, v); else *o = v;
            } else if (a.halo != nullptr) {
              const int hl = a.o_padL + a.o_padR;
              float* o = nullptr;
              if (ii < 0) o = a.halo + rowoff * hl + P;
              else if (ii - a.Lout < a.o_padR) o = a.halo + rowoff * hl + a.o_padL + (ii - a.Lout);
              if (o) { if (partial) atomicAdd(o, v); else *o = v; }
            }
          }
        }
      }
    }
  }
  }  // segment loop
}

// ====================================================================================
// packing: w[m][n][K] fp32 -> NPL bf16 planes in the kernel's tile order
// ====================================================================================
// F: piece (cg,u,g,row) holds channels cv = 16cg+8g+e = (n,r): w[row][n][S*u+r]
// T: piece (cg,u',g,row=(r,nn)) holds channels m = 16cg+8g+e: w[m][nn][S*(U-1-u')+rho(r)]
//
// Both are transposes through LDS so that the global reads run along the K taps of consecutive
// (m, n) rows and the 16-byte piece writes along consecutive rows (round 1's one-thread-per-piece
// kernel read 8 floats at a stride of N*K per lane: 0.96 ms per step for 1.1 GB of traffic).
//
// F form: one block = 64 rows m x one half-group g (8 virtual channels = 8/S real channels),
// all U taps: reads w[m][n0 .. n0 + 8/S)[0..K) (contiguous per row), writes U x 64 pieces.
__global__ __launch_bounds__(256) void pack_bf_f_kernel(const float* __restrict__ w,
                                                        __bf16* __restrict__ out, long plane_stride,
                                                        int planes, int M, int N, int K, int S, int U,
                                                        int RP) {
  extern __shared__ float tf[];                    // [RT rows][W + 1]: (channel in group, tap)
  const int hg = blockIdx.y;                       // 2*cg + g
  const int NC = 8 / S;                            // real channels of this half-group
  const int n0 = hg * NC;
  const int tid = threadIdx.x;
  const int W = NC * 32;                           // floats staged per row (taps padded to 32)
  const int RT = S == 1 ? 32 : 64;                 // rows per block (LDS: RT * (W + 1) floats)
  const int m0 = blockIdx.x * RT;
  auto t = [&](int ml, int x) -> float& { return tf[ml * (W + 1) + x]; };
  for (int e = tid; e < RT * W; e += 256) {
    const int ml = e / W, x = e - ml * W;
    const int c = x >> 5, k = x & 31;
    const int m = m0 + ml, n = n0 + c;
    t(ml, x) = (m < M && n < N && k < K) ? w[((size_t)m * N + n) * K + k] : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < U * RT; e += 256) {
    const int u = e / RT, ml = e - u * RT;
    if (m0 + ml >= RP) continue;
    bf16x8 pl[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = q / S, r = q % S;              // virtual channel q of the half-group = (c, r)
      __bf16 p1, p2, p3;
      split3(t(ml, c * 32 + S * u + r), p1, p2, p3);
      pl[0][q] = p1; pl[1][q] = p2; pl[2][q] = p3;
    }
    const long pc = ((long)((hg >> 1) * U + u) * 2 + (hg & 1)) * RP + m0 + ml;
    for (int p = 0; p < planes; ++p)
      *reinterpret_cast<u32x4*>(out + p * plane_stride + pc * 8) = __builtin_bit_cast(u32x4, pl[p]);
  }
}

// T form: one block = 8 channels m (one half-group) x 64 output channels nn, all taps: reads
// w[m][nn0 .. nn0+64)[0..K) (one contiguous run per m), writes (U taps x S phases) x 64 pieces.
__global__ __launch_bounds__(256) void pack_bf_t_kernel(const float* __restrict__ w,
                                                        __bf16* __restrict__ out, long plane_stride,
                                                        int planes, int M, int N, int K, int S, int U,
                                                        int RP, int NP, int pad) {
  __shared__ float t[8][32 * 33];                 // [m][nn (pitch 33)][tap]
  const int hg = blockIdx.y;
  const int nn0 = blockIdx.x * 32;
  const int tid = threadIdx.x;
  for (int e = tid; e < 8 * 32 * 32; e += 256) {
    const int ml = e >> 10, x = e & 1023;
    const int nl = x >> 5, k = x & 31;
    const int m = 8 * hg + ml, nn = nn0 + nl;
    t[ml][nl * 33 + k] = (m < M && nn < N && k < K) ? w[((size_t)m * N + nn) * K + k] : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < U * S * 32; e += 256) {
    const int nl = e & 31;
    const int ur = e >> 5;                        // u' * S + r
    const int up = ur / S, r = ur - up * S;
    const int nn = nn0 + nl;
    if (nn >= NP) continue;
    const int rho = (r + pad) % S;
    const int k = S * (U - 1 - up) + rho;
    bf16x8 pl[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __bf16 p1, p2, p3;
      split3(t[q][nl * 33 + k], p1, p2, p3);
      pl[0][q] = p1; pl[1][q] = p2; pl[2][q] = p3;
    }
    const long pc = ((long)((hg >> 1) * U + up) * 2 + (hg & 1)) * RP + (long)r * NP + nn;
    for (int p = 0; p < planes; ++p)
      *reinterpret_cast<u32x4*>(out + p * plane_stride + pc * 8) = __builtin_bit_cast(u32x4, pl[p]);
  }
}

static inline int bf_f_pitch(int M) { return round_up(M, 128); }

extern "C" size_t segan_packed_bf_bytes(int M, int N, int S, int tform, int planes) {
  if (!(S == 1 || S == 2 || S == 4) || M <= 0 || N <= 0 || planes < 1 || planes > 3) return 0;
  const int U = 32 / S;
  if (!tform) {
    const int ng = ceil_div(N * S, 16);
    return (size_t)planes * ng * U * 2 * bf_f_pitch(M) * 8 * sizeof(__bf16);
  }
  const int ng = ceil_div(M, 16);
  return (size_t)planes * ng * U * 2 * (S * t_np(N, S)) * 8 * sizeof(__bf16);
}

extern "C" int segan_pack_weights_bf(const float* w, void* out, int M, int N, int K, int S,
                                     int tform, int pad_t, int planes, void* stream) {
  SEGAN_REQUIRE(w && out, "pack_weights_bf: NULL pointer");
  SEGAN_REQUIRE(S == 1 || S == 2 || S == 4, "pack_weights_bf: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32 && M > 0 && N > 0, "pack_weights_bf: bad sizes");
  SEGAN_REQUIRE(planes == 1 || planes == 3, "pack_weights_bf: planes must be 1 or 3");
  const int U = 32 / S;
  const int NP = t_np(N, S);
  const int RP = tform ? S * NP : bf_f_pitch(M);
  const int ng = tform ? ceil_div(M, 16) : ceil_div(N * S, 16);
  const long npieces = (long)ng * U * 2 * RP;
  const long plane_stride = npieces * 8;
  hipStream_t st = (hipStream_t)stream;
  if (!tform) {
    const int RT = S == 1 ? 32 : 64;
    const size_t lds = (size_t)RT * ((8 / S) * 32 + 1) * sizeof(float);
    hipLaunchKernelGGL(pack_bf_f_kernel, dim3(RP / RT, 2 * ng), dim3(256), lds, st, w, (__bf16*)out,
                       plane_stride, planes, M, N, K, S, U, RP);
  } else {
    hipLaunchKernelGGL(pack_bf_t_kernel, dim3(ceil_div(NP, 32), 2 * ng), dim3(256), 0, st, w,
                       (__bf16*)out, plane_stride, planes, M, N, K, S, U, RP, NP, pad_t);
  }
  return segan_check_launch("pack_weights_bf");
}

// ====================================================================================
// launchers
// ====================================================================================
template <int NB, int WM, int U, bool IN_HI, bool OUT_HI, int SHIFTMASK, int NPL>
static int launch_bf(CorrArgs a, BfExtra x, hipStream_t st) {
  constexpr int MB = 128;
  constexpr int S = 32 / U;
  constexpr int TU = (NPL == 3) ? 2 : (U >= 8 ? 4 : U);
  constexpr int TCH = U / TU;
  const size_t lds = (size_t)(2 * NPL * TU * 2 * MB + NPL * 2 * a.RLs) * 16;
  if (lds > 160 * 1024 || 2 * a.RLs > 512) {
    segan_set_error("corr_bf: tile does not fit (RLs=%d)", a.RLs);
    return SEGAN_EUNSUPPORTED;
  }
  auto kern = corr_bf_kernel<MB, NB, WM, U, IN_HI, OUT_HI, SHIFTMASK, NPL, TU>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int nrowtiles = OUT_HI ? a.NP / (MB / S) : ceil_div(a.Rvalid, MB);
  a.rt0 = (!OUT_HI && a.out0 == nullptr) ? a.OC0 / MB : 0;
  const int ntiles = (nrowtiles - a.rt0) * a.ncoltiles;
  const int nst = x.ngroups * TCH;
  a.sk_nfull = ntiles;
  a.sk_units = 0;
  a.sk_total = 0;
  unsigned grid = (unsigned)ntiles;
  const double classic_eff = (double)ntiles / (256.0 * ceil_div(ntiles, 256));
  if (a.act == SEGAN_ACT_NONE && ntiles >= 64 && nst >= 8 && classic_eff < 0.97) {
    static int occ_cache = 0;
    static size_t occ_lds = 0;
    if (occ_cache == 0 || occ_lds != lds) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern),
                                                       256, lds) != hipSuccess || nb < 1)
        nb = 1;
      occ_cache = nb > 4 ? 4 : nb;
      occ_lds = lds;
    }
    const int G = 256 * occ_cache;
    a.sk_nfull = (ntiles / G) * G;
    a.sk_total = (long)(ntiles - a.sk_nfull) * nst;
    a.sk_units = (int)((a.sk_total + G - 1) / G);
    grid = (unsigned)G;
    hipError_t e = hipSuccess;
    if (a.out0 && a.out0_elems) e = hipMemsetAsync(a.out0, 0, a.out0_elems * sizeof(float), st);
    if (e == hipSuccess && a.out1 && a.out1_elems)
      e = hipMemsetAsync(a.out1, 0, a.out1_elems * sizeof(float), st);
    if (e == hipSuccess && a.halo && a.halo_elems)
      e = hipMemsetAsync(a.halo, 0, a.halo_elems * sizeof(float), st);
    if (e != hipSuccess) {
      segan_set_error("corr_bf: memset failed: %s", hipGetErrorString(e));
      return SEGAN_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, x);
  return segan_check_launch("corr_bf_kernel");
}

static int prep(CorrArgs& a, hipStream_t st) {
  constexpr int NB = 128;
  a.ncoltiles = ceil_div(a.Ctot, NB);
  a.RLs = NB + samples_per_tile(a.Tcols, NB) * a.H;
  a.in_identity = (!a.in.scale && !a.in.shift && !a.in.slope) ? 1 : 0;
  if (int e = segan_src_defaults(&a.in, st, "corr_bf")) return e;
  if ((long)a.B * (a.in.C0 + a.in.C1) * a.Lin >= (1L << 30) || a.Lin >= (1 << 24) ||
      a.in.C0 + a.in.C1 >= (1 << 24)) {
    segan_set_error("corr_bf: input exceeds the 2^30 element / 2^24 row indexing limits");
    return SEGAN_EUNSUPPORTED;
  }
  return SEGAN_OK;
}

int segan_corr_bf_f(CorrArgs& a, int U, const void* wp3, int planes, hipStream_t st) {
  if (a.Rvalid <= 64) {
    segan_set_error("corr_bf: layers with <= 64 output rows stay on the fp32 kernel");
    return SEGAN_EUNSUPPORTED;
  }
  if (int e = prep(a, st)) return e;
  BfExtra x;
  x.wp3 = (const __bf16*)wp3;
  x.ngroups = ceil_div(a.Cv, 16);
  a.RP = bf_f_pitch(a.Rvalid);
  x.plane_stride = (long)x.ngroups * U * 2 * a.RP * 8;
  if (U == 8) return planes == 3 ? launch_bf<128, 2, 8, true, false, 0, 3>(a, x, st)
                                 : launch_bf<128, 2, 8, true, false, 0, 1>(a, x, st);
  if (U == 16) return planes == 3 ? launch_bf<128, 2, 16, true, false, 0, 3>(a, x, st)
                                  : launch_bf<128, 2, 16, true, false, 0, 1>(a, x, st);
  segan_set_error("corr_bf: stride 1 is not implemented on the bf16 path");
  return SEGAN_EUNSUPPORTED;
}

int segan_corr_bf_t(CorrArgs& a, int U, const void* wp3, int planes, hipStream_t st) {
  if (int e = prep(a, st)) return e;
  BfExtra x;
  x.wp3 = (const __bf16*)wp3;
  x.ngroups = ceil_div(a.Cv, 16);
  // a.RP = S*NP already (t_pitch)
  x.plane_stride = (long)x.ngroups * U * 2 * a.RP * 8;
  const int mask = (a.rowshift[0] ? 1 : 0) | (a.rowshift[1] ? 2 : 0) | (a.rowshift[2] ? 4 : 0) |
                   (a.rowshift[3] ? 8 : 0);
  if (U == 8 && mask == 8)
    return planes == 3 ? launch_bf<128, 1, 8, false, true, 8, 3>(a, x, st)
                       : launch_bf<128, 1, 8, false, true, 8, 1>(a, x, st);
  if (U == 8 && mask == 0)
    return planes == 3 ? launch_bf<128, 1, 8, false, true, 0, 3>(a, x, st)
                       : launch_bf<128, 1, 8, false, true, 0, 1>(a, x, st);
  if (U == 16 && mask == 0)
    return planes == 3 ? launch_bf<128, 1, 16, false, true, 0, 3>(a, x, st)
                       : launch_bf<128, 1, 16, false, true, 0, 1>(a, x, st);
  segan_set_error("corr_bf: unsupported T-form geometry (U=%d, shift mask %d)", U, mask);
  return SEGAN_EUNSUPPORTED;
}
