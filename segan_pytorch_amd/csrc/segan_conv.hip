// segan_conv.hip — the two contraction kernels of the SEGAN GAN step for gfx950.
//
// Every strided conv / transposed conv of the generator and discriminator
// (reference segan/models/modules.py:75-141), forward and backward, is one of three
// forms over the polyphase split  k = S*u + r  of the K<=32 taps (see
// segan_pytorch_amd/layout.py, which restates this arithmetic for the CPU tests):
//
//   corr<IN_HI=1,OUT_HI=0>  "F": out[m,t]      = sum_{(n,r),u} Wf[(n,r),u,m] * X_r[n,t+u]
//        conv fwd (modules.py:99) and deconv dgrad
//   corr<IN_HI=0,OUT_HI=1>  "T": y[n,S*q+r]   = sum_{m,u'} Wt[m,u',(r,n)] * x[m,q+c(r)-(U-1)+u']
//        deconv fwd (modules.py:136) and conv dgrad
//   wgrad                   "W": dW[m,n,S*u+r] += sum_{b,t} lo[b,m,t] * HI_r[b,n,t+u]
//
// All three are exact-fp32 implicit GEMMs on v_mfma_f32_32x32x2_f32 (bitwise an fmaf
// chain): 256-thread workgroups, 2x2 waves, each wave a (MB/2)x(NB/2) tile of 32x32
// MFMA blocks, operands staged through LDS.  Nothing is im2col'ed: the activation
// tile in LDS is the raw (phase-split) signal with a U-1 halo per sample, and the
// 8/16/32 taps of a phase read it at shifted addresses.  Reflect padding, the
// discriminator's circular phase shift, both torch.cat's, the alpha skip scale,
// BatchNorm-normalise and PReLU are all applied while the tile is staged
// (segan_src), so none of those tensors is ever materialised in HBM.
#include "segan_conv_shared.h"

// Staging discipline (both kernels): load_chunk() only ISSUES global loads — every
// address is clamped to a valid element, so there is no branch and no wait between
// them and they stay in flight under the MFMA loop; masking, the segan_src transform
// and the LDS writes happen in store_chunk(), after the compute of the previous chunk.
//
// Tile geometry: MB rows x NB columns per 256-thread workgroup, waves WM x (4/WM).
//   F form: rows = output channels m; waves 2x2.
//   T form: rows = (phase r, channel n) with ALL S phases of MB/S channels in one tile and
//           waves 1x4 (NB=128) so that one lane ends up holding the S consecutive output
//           samples S*q..S*q+S-1 of a channel: full-line stores instead of stride-S ones.
template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, bool SHIFT, int MAXPOS, int KC>
__global__ __launch_bounds__(256, 2) void corr_kernel(const CorrArgs a) {
  constexpr int S = 32 / U;
  constexpr int SI = IN_HI ? S : 1;   // indices per staged position
  constexpr int CV = KC / U;          // virtual channels per chunk
  constexpr int WN = 4 / WM;
  constexpr int NI = MB / (32 * WM);
  constexpr int NJ = NB / (32 * WN);
  constexpr int NPT = MB / S;         // T form: channels per tile
  static_assert(!OUT_HI || NPT % 32 == 0, "T-form tiles hold whole 32-row phase blocks");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RLs = a.RLs;
  float* Wl0 = smem;                  // [2][KC*MB]
  float* Il0 = smem + 2 * KC * MB;   // [2][CV*RLs]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  // Work decomposition (data-parallel + stream-K hybrid).  The first a.sk_nfull tiles are
  // whole-tile work items, strided over the grid.  The remaining tiles — fewer than the
  // grid, i.e. the partial last round that would leave CUs idle — are cut in the (tile,
  // chunk) iteration space into equal contiguous ranges, one per workgroup; a tile cut
  // across workgroups is combined with fp32 atomics into the zero-initialised output (its
  // bias is added by the piece that holds chunk 0).  Classic launch: sk_nfull = #tiles.
  const int nch = (a.Ktot + KC - 1) / KC;
  int tileA = blockIdx.x;
  long unit = (long)blockIdx.x * a.sk_units;
  const long unit_end = min(unit + (long)a.sk_units, a.sk_total);
  for (;;) {
  int tile, c0, c1;
  if (tileA < a.sk_nfull) {
    tile = tileA; c0 = 0; c1 = nch;
    tileA += gridDim.x;
  } else if (unit < unit_end) {
    const int t = (int)(unit / nch);
    c0 = (int)(unit - (long)t * nch);
    c1 = min(nch, c0 + (int)(unit_end - unit));
    unit += c1 - c0;
    tile = a.sk_nfull + t;
  } else {
    break;
  }
  const bool partial = (c0 != 0) || (c1 != nch);
  const int rowtile = a.rt0 + tile / a.ncoltiles;
  const int coltile = tile % a.ncoltiles;
  const int m0 = rowtile * MB;          // F form: first row; T form: n0 = rowtile * NPT
  const int n0 = rowtile * NPT;
  if (!OUT_HI) {
    // dual destination: skip tiles whose rows all go to a NULL destination
    if (a.out0 == nullptr && m0 + MB <= a.OC0) continue;
    if (a.out1 == nullptr && m0 >= a.OC0) continue;
  }
  const ColTile ct = make_coltile(coltile * NB, a.Tcols, NB);

  // ---- per-thread staging positions of the activation tile (fixed for all chunks) ----
  int pos_off[MAXPOS][SI];
  unsigned pos_ok[MAXPOS];
  int pos_bo0[MAXPOS], pos_bo1[MAXPOS];
#pragma unroll
  for (int pp = 0; pp < MAXPOS; ++pp) {
    const int j = tid + 256 * pp;
    pos_ok[pp] = 0u;
    pos_bo0[pp] = 0;
    pos_bo1[pp] = 0;
#pragma unroll
    for (int r = 0; r < SI; ++r) pos_off[pp][r] = 0;
    if (j < RLs) {
      int s, tau;
      lds_pos_decode(ct, j, a.Tcols, a.H, s, tau);
      const int b = ct.b0 + s;
      if (b < a.B) {
        pos_bo0[pp] = b * a.in.C0 * a.Lin;
        pos_bo1[pp] = b * a.in.C1 * a.Lin;
        const int wq = tau + a.win_start;
        if (IN_HI) {
#pragma unroll
          for (int r = 0; r < SI; ++r) {
            const int idx = segan_hi_index(S * wq + r, a.Lin, a.padL, a.mode, a.roll);
            if (idx >= 0) { pos_off[pp][r] = idx; pos_ok[pp] |= 1u << r; }
          }
        } else if (wq >= 0 && wq < a.Lin) {
          pos_off[pp][0] = wq;
          pos_ok[pp] = 1u;
        }
      }
    }
  }

  // ---- per-lane operand offsets ----
  int aoff[NI], boff[NJ], rsh[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rloc = 32 * (wm * NI + i);
    aoff[i] = h * MB + rloc + l31;
    rsh[i] = SHIFT ? a.rowshift[rloc / NPT] : 0;
  }
  int col_b[NJ], col_t[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cl = wn * (NB / WN) + 32 * j + l31;
    const int col = ct.col0 + cl;
    if (col < a.Ctot) {
      const int b = col / a.Tcols;
      col_b[j] = b;
      col_t[j] = col - b * a.Tcols;
      boff[j] = cl + (b - ct.b0) * a.H + h;
    } else {
      col_b[j] = -1;
      col_t[j] = 0;
      boff[j] = h;
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
  constexpr int F4R = MB / 4;        // float4 per weight row
  constexpr int RPP = 256 / F4R;     // rows per pass
  constexpr int NPASS = KC / RPP;
  f32x4 wreg[NPASS];
  float ireg[CV][MAXPOS];
  const int wrow = tid / F4R, wc4 = tid % F4R;
  // global column of LDS column 4*wc4: F form m0 + c; T form phase-major (r*NP + n0 + nl)
  const int wgcol = OUT_HI ? ((4 * wc4) / NPT) * a.NP + n0 + (4 * wc4) % NPT : m0 + 4 * wc4;
  const float* wbase = a.wp + (size_t)wrow * a.RP + wgcol;

  auto load_chunk = [&](int ch) {
    const float* wsrc = wbase + (size_t)(ch * KC) * a.RP;
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
      wreg[p] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)(RPP * p) * a.RP);
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      int cv = ch * CV + c;
      cv = cv < a.Cv ? cv : 0;
      const int n = IN_HI ? cv / S : cv;
      const int r = IN_HI ? c % S : 0;  // CV is a multiple of S
      const bool seg1 = n >= a.in.C0;
      const float* rowp = seg1 ? a.in.p1 + (size_t)(n - a.in.C0) * a.Lin
                               : a.in.p0 + (size_t)n * a.Lin;
#pragma unroll
      for (int pp = 0; pp < MAXPOS; ++pp)
        ireg[c][pp] = rowp[(seg1 ? pos_bo1[pp] : pos_bo0[pp]) + pos_off[pp][r]];
    }
  };
  auto store_chunk = [&](int ch, int buf) {
    float* Wl = Wl0 + buf * (KC * MB);
    float* Il = Il0 + buf * (CV * RLs);
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
      *reinterpret_cast<f32x4*>(Wl + (wrow + RPP * p) * MB + 4 * wc4) = wreg[p];
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      const int cv = ch * CV + c;
      const bool cvalid = cv < a.Cv;
      const int n = IN_HI ? cv / S : cv;
      const int r = IN_HI ? c % S : 0;
      const ChanXf xf = segan_chan_xf(a.in, cvalid ? n : 0);
#pragma unroll
      for (int pp = 0; pp < MAXPOS; ++pp) {
        const int j = tid + 256 * pp;
        const bool ok = cvalid && ((pos_ok[pp] >> r) & 1u);
        const float v = ok ? segan_apply_xf(xf, ireg[c][pp]) : 0.0f;
        if (j < RLs) Il[c * RLs + j] = v;
      }
    }
  };

  load_chunk(c0);
  store_chunk(c0, 0);
  __syncthreads();
  for (int ch = c0; ch < c1; ++ch) {
    const int buf = (ch - c0) & 1;
    if (ch + 1 < c1) load_chunk(ch + 1);
    const float* Wl = Wl0 + buf * (KC * MB);
    const float* Il = Il0 + buf * (CV * RLs);
    // operands of step s+1 are read from LDS before the MFMAs of step s are issued
    // (two named register sets; everything is unrolled so all indices are static)
    constexpr int NBI = SHIFT ? NI : 1;
    float av0[NI], av1[NI], bv0[NBI][NJ], bv1[NBI][NJ];
    auto read_step = [&](int s, float (&av)[NI], float (&bv)[NBI][NJ]) {
      const int kk = 2 * s;
      const int c = kk / U, u = kk % U;
      const float* wr = Wl + kk * MB;
      const float* ir = Il + c * RLs + u;
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i] = wr[aoff[i]];
#pragma unroll
      for (int i = 0; i < NBI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[i][j] = ir[boff[j] + (SHIFT ? rsh[i] : 0)];
    };
    auto mma_step = [&](const float (&av)[NI], const float (&bv)[NBI][NJ]) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[SHIFT ? i : 0][j], acc[i][j],
                                                           0, 0, 0);
    };
    // sched_barrier pins "reads of step s+1, then MFMAs of step s" so the LDS latency of the
    // next operands is covered by the MFMAs instead of being exposed
    read_step(0, av0, bv0);
#pragma unroll
    for (int s = 0; s < KC / 2; s += 2) {
      read_step(s + 1, av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < KC / 2) read_step(s + 2, av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch + 1 < c1) store_chunk(ch + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  const bool add_bias = (c0 == 0);
  if (!OUT_HI) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + 32 * (wm * NI + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= a.Rvalid) continue;
        // LO store: out[b, row, t]
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
    // HI store.  Row block ib of the tile is phase r = 32*ib / NPT of channels n0 + nl.
    constexpr bool QUAD = (S == 4 && WM == 1 && NI == 4);  // lane holds all 4 phases of (n, q)
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

// fold the reflect halo of a conv dgrad back into dx.  When L > padL + padR + 1 the padL left
// and padR right halo samples mirror onto distinct elements of a row, so one thread per
// (row, halo sample) is race-free; shorter rows fall back to one thread per row.
__global__ void fold_halo_kernel(float* dx, const float* halo, int rows, int L, int padL,
                                 int padR, int roll, int per_sample) {
  const int hl = padL + padR;
  if (per_sample) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * hl) return;
    const int row = (int)(t / hl), j = (int)(t - (long)row * hl);
    const int P = j < padL ? j : L + j;       // right halo sample j-padL sits at L + padL + (j-padL)
    const int idx = segan_hi_index(P, L, padL, SEGAN_PAD_REFLECT, roll);
    if (idx >= 0) dx[(size_t)row * L + idx] += halo[(size_t)row * hl + j];
    return;
  }
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  float* d = dx + (size_t)row * L;
  const float* hrow = halo + (size_t)row * hl;
  for (int P = 0; P < padL; ++P) {
    const int idx = segan_hi_index(P, L, padL, SEGAN_PAD_REFLECT, roll);
    if (idx >= 0) d[idx] += hrow[P];
  }
  for (int P2 = 0; P2 < padR; ++P2) {
    const int idx = segan_hi_index(L + padL + P2, L, padL, SEGAN_PAD_REFLECT, roll);
    if (idx >= 0) d[idx] += hrow[padL + P2];
  }
}

// packed-weight geometry (shared by the pack kernels and the launchers)
static inline int f_pitch(int M) { return M <= 64 ? 64 : round_up(M, 128); }
static inline int f_rows(int N) { return round_up(N * 32, KCH); }
static inline int t_pitch(int N, int S) { return S * t_np(N, S); }
static inline int t_rows(int M, int S) { return round_up(M * (32 / S), KCH); }

static bool streamk_enabled() {
  static const int env = [] { const char* e = getenv("SEGAN_STREAMK"); return e ? atoi(e) : 1; }();
  return env != 0;
}

template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, bool SHIFT, int MAXPOS, int KC>
static int launch_corr_t(CorrArgs a, hipStream_t st, bool allow_sk) {
  constexpr int CV = KC / U;
  constexpr int S = 32 / U;
  const int nrowtiles = OUT_HI ? a.NP / (MB / S) : ceil_div(a.Rvalid, MB);
  const size_t lds = (size_t)(2 * KC * MB + 2 * CV * a.RLs) * sizeof(float);
  if (lds > 160 * 1024) {
    segan_set_error("corr: LDS tile %zu B too large (RLs=%d)", lds, a.RLs);
    return SEGAN_EUNSUPPORTED;
  }
  auto kern = corr_kernel<MB, NB, WM, U, IN_HI, OUT_HI, SHIFT, MAXPOS, KC>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  // rows below rt0 all go to a NULL destination (the z half of the first decoder layer)
  a.rt0 = (!OUT_HI && a.out0 == nullptr) ? a.OC0 / MB : 0;
  const int ntiles = (nrowtiles - a.rt0) * a.ncoltiles;
  const int nch = ceil_div(a.Ktot, KC);
  a.sk_nfull = ntiles;
  a.sk_units = 0;
  a.sk_total = 0;
  unsigned grid = (unsigned)ntiles;
  // hybrid when one-tile-per-workgroup would leave >3 % of the CU-time idle at the end
  const double classic_eff = (double)ntiles / (256.0 * ceil_div(ntiles, 256));
  if (allow_sk && streamk_enabled() && a.act == SEGAN_ACT_NONE && ntiles >= 64 && nch >= 8 &&
      classic_eff < 0.97) {
    // workgroups resident per CU (registers / LDS): the grid is exactly one resident round
    static int occ_cache = 0;
    static size_t occ_lds = 0;
    if (occ_cache == 0 || occ_lds != lds) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern),
                                                       256, lds) != hipSuccess || nb < 1)
        nb = 2;
      occ_cache = nb > 4 ? 4 : nb;
      occ_lds = lds;
    }
    const int occ = occ_cache;
    const int G = 256 * occ;
    a.sk_nfull = (ntiles / G) * G;
    const int rem = ntiles - a.sk_nfull;
    a.sk_total = (long)rem * nch;
    a.sk_units = (int)((a.sk_total + G - 1) / G);
    grid = (unsigned)G;
    // tiles cut across workgroups are accumulated with atomics: zero the destinations
    hipError_t e = hipSuccess;
    if (a.out0 && a.out0_elems) e = hipMemsetAsync(a.out0, 0, a.out0_elems * sizeof(float), st);
    if (e == hipSuccess && a.out1 && a.out1_elems)
      e = hipMemsetAsync(a.out1, 0, a.out1_elems * sizeof(float), st);
    if (e == hipSuccess && a.halo && a.halo_elems)
      e = hipMemsetAsync(a.halo, 0, a.halo_elems * sizeof(float), st);
    if (e != hipSuccess) {
      segan_set_error("corr: memset failed: %s", hipGetErrorString(e));
      return SEGAN_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  return segan_check_launch("corr_kernel");
}

// All workgroups of a launch do the same amount of work and several are resident per CU
// sharing its MFMA pipes, so a launch takes ceil(blocks / 256) block times; the half-size
// tile is used when that quantisation is better (the deep layers: few, long tiles).
static bool prefer_half_tile(int nb_full, int nb_half) {
  static const int env = [] { const char* e = getenv("SEGAN_CORR_HALF"); return e ? atoi(e) : -1; }();
  if (env == 0) return false;
  if (env == 1) return true;
  const double t_full = (double)ceil_div(nb_full, 256);
  const double t_half = 0.5 * 1.06 * (double)ceil_div(nb_half, 256);   // 6 % tile-size penalty
  return t_half < t_full;
}

// ---- F form (conv forward, deconv data gradient) ----
template <int U>
static int launch_corr_f(CorrArgs& a, hipStream_t st) {
  constexpr int NB = 128;
  a.ncoltiles = ceil_div(a.Ctot, NB);
  a.RLs = NB + samples_per_tile(a.Tcols, NB) * a.H;
  bool small = a.Rvalid <= 64;
  if (!small && !streamk_enabled())
    small = prefer_half_tile(ceil_div(a.Rvalid, 128) * a.ncoltiles, ceil_div(a.Rvalid, 64) * a.ncoltiles);
  if (a.RLs <= 256) {
    if (!small && U <= 16)
      return launch_corr_t<128, NB, 2, U, true, false, false, 1, 32>(a, st, true);
    return small ? launch_corr_t<64, NB, 2, U, true, false, false, 1, KCH>(a, st, false)
                 : launch_corr_t<128, NB, 2, U, true, false, false, 1, KCH>(a, st, true);
  }
  if (a.RLs > 512) {
    segan_set_error("corr: sample length %d too short for stride %d (RLs=%d)", a.Tcols, 32 / U,
                    a.RLs);
    return SEGAN_EUNSUPPORTED;
  }
  return small ? launch_corr_t<64, NB, 2, U, true, false, false, 2, KCH>(a, st, false)
               : launch_corr_t<128, NB, 2, U, true, false, false, 2, KCH>(a, st, true);
}

// ---- T form (deconv forward, conv data gradient) ----
template <int U, bool SHIFT>
static int launch_corr_tt(CorrArgs& a, hipStream_t st) {
  constexpr int S = 32 / U;
  const int nrt = a.NP / (128 / S);
  const int ct128 = ceil_div(a.Ctot, 128), ct64 = ceil_div(a.Ctot, 64);
  const bool half = !streamk_enabled() && prefer_half_tile(nrt * ct128, nrt * ct64);
  const int NB = half ? 64 : 128;
  a.ncoltiles = half ? ct64 : ct128;
  a.RLs = NB + samples_per_tile(a.Tcols, NB) * a.H;
  if (a.RLs > 512) {
    segan_set_error("corr: sample length %d too short for stride %d (RLs=%d)", a.Tcols, 32 / U,
                    a.RLs);
    return SEGAN_EUNSUPPORTED;
  }
  constexpr int KC = U <= 16 ? 32 : KCH;
  if (a.RLs <= 256)
    return half ? launch_corr_t<128, 64, 2, U, false, true, SHIFT, 1, KC>(a, st, false)
                : launch_corr_t<128, 128, 1, U, false, true, SHIFT, 1, KC>(a, st, true);
  return half ? launch_corr_t<128, 64, 2, U, false, true, SHIFT, 2, KCH>(a, st, false)
              : launch_corr_t<128, 128, 1, U, false, true, SHIFT, 2, KCH>(a, st, true);
}

template <bool IN_HI, bool OUT_HI>
static int launch_corr(CorrArgs& a, int U, hipStream_t st) {
  static const int prio_env = [] { const char* e = getenv("SEGAN_PRIO"); return e ? atoi(e) : 0; }();
  a.prio_mode = prio_env;
  if (int e = segan_src_defaults(&a.in, st, "corr")) return e;
  const long in_elems = (long)a.B * (a.in.C0 + a.in.C1) * a.Lin;
  if (in_elems >= (1L << 31)) {
    segan_set_error("corr: input of %ld elements exceeds the 2^31 indexing limit", in_elems);
    return SEGAN_EUNSUPPORTED;
  }
  if (!OUT_HI) {
    switch (U) {
      case 8: return launch_corr_f<8>(a, st);
      case 16: return launch_corr_f<16>(a, st);
      case 32: return launch_corr_f<32>(a, st);
    }
  } else {
    const bool shift = a.rowshift[0] | a.rowshift[1] | a.rowshift[2] | a.rowshift[3];
    switch (U) {
      case 8: return shift ? launch_corr_tt<8, true>(a, st) : launch_corr_tt<8, false>(a, st);
      case 16: return shift ? launch_corr_tt<16, true>(a, st) : launch_corr_tt<16, false>(a, st);
      case 32: return shift ? launch_corr_tt<32, true>(a, st) : launch_corr_tt<32, false>(a, st);
    }
  }
  segan_set_error("corr: unsupported stride (U=%d)", U);
  return SEGAN_EUNSUPPORTED;
}

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
  for (int mc0 = 0; mc0 < M; mc0 += MC) {
#pragma unroll 8
    for (int mc = 0; mc < MC; ++mc) {
      const int m = mc0 + mc < M ? mc0 + mc : 0;
      const bool seg1 = m >= a.in.C0;
      const float* rowp = seg1 ? a.in.p1 + (size_t)(m - a.in.C0) * a.Lin + bo1
                               : a.in.p0 + (size_t)m * a.Lin + bo0;
      const ChanXf xf = segan_chan_xf(a.in, m);
      const float v0 = rowp[o0], v1 = rowp[o1];
      const bool mok = mc0 + mc < M;
      xs[mc][tid] = (mok && ok0) ? segan_apply_xf(xf, v0) : 0.0f;
      if (tid < U) xs[mc][256 + tid] = (mok && ok1) ? segan_apply_xf(xf, v1) : 0.0f;
    }
    __syncthreads();
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

// `a` is filled exactly as for the MFMA T form; w is the UNPACKED weight [M][N][K]
static int launch_tsmall(CorrArgs& a, const float* w, int K, int M, int N, int S, int pad,
                         hipStream_t st) {
  if (int e = segan_src_defaults(&a.in, st, "tsmall")) return e;
  if (N == 1) {
    if (S == 4) return launch_tsmall_sn<4, 1>(a, w, K, M, pad, st);
    if (S == 2) return launch_tsmall_sn<2, 1>(a, w, K, M, pad, st);
    return launch_tsmall_sn<1, 1>(a, w, K, M, pad, st);
  }
  if (S == 4) return launch_tsmall_sn<4, 2>(a, w, K, M, pad, st);
  if (S == 2) return launch_tsmall_sn<2, 2>(a, w, K, M, pad, st);
  return launch_tsmall_sn<1, 2>(a, w, K, M, pad, st);
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
  constexpr int U = 32 / S;
  constexpr int XW = S * 256 + 32;
  constexpr int WST = N * 32 + 4;          // row stride of the weight copy (16-B aligned)
  __shared__ __attribute__((aligned(16))) float xs[N][XW];
  __shared__ __attribute__((aligned(16))) float ws[64 * WST];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 256;
  for (int j = tid; j < XW; j += 256) {
    const int idx = segan_hi_index(S * t0 + j, a.Lin, a.padL, a.mode, a.roll);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float v = 0.0f;
      if (idx >= 0) v = segan_apply_xf(segan_chan_xf(a.in, n), segan_src_row(a.in, b, n, a.Lin)[idx]);
      xs[n][j] = v;
    }
  }
  const int t = t0 + tid;
  for (int m0 = 0; m0 < M; m0 += 64) {
    // packed F layout: w[m][n][S*u + r] = wp[((n*S + r)*U + u) * RP + m]; rows of taps >= K
    // are zero.  Lanes run along m so the global reads are coalesced.
    for (int e = tid; e < 64 * N * 32; e += 256) {
      const int ml = e & 63, nk = e >> 6;
      const int n = nk >> 5, k = nk & 31;
      const int row = (n * S + k % S) * U + k / S;
      const int m = m0 + ml;
      ws[ml * WST + n * 32 + k] = m < a.RP ? a.wp[(size_t)row * a.RP + m] : 0.0f;
    }
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

static int launch_fsmall(CorrArgs& a, int M, int N, int S, hipStream_t st) {
  if (int e = segan_src_defaults(&a.in, st, "fsmall")) return e;
  dim3 grid(ceil_div(a.Lout, 256), a.B);
#define FS(SS, NN) hipLaunchKernelGGL((fsmall_kernel<SS, NN>), grid, dim3(256), 0, st, a, M)
  if (N == 1) { if (S == 4) FS(4, 1); else if (S == 2) FS(2, 1); else FS(1, 1); }
  else { if (S == 4) FS(4, 2); else if (S == 2) FS(2, 2); else FS(1, 2); }
#undef FS
  return segan_check_launch("fsmall_kernel");
}

// ====================================================================================
// wgrad kernel
// ====================================================================================

// x / Ls for 0 <= x < Ls + TK (Ls >= TK: one compare; else exact multiply-shift, x < 64)
template <int TK>
__device__ __forceinline__ int wg_sdiv(int x, int Ls, int magic) {
  return (Ls >= TK) ? (x >= Ls ? 1 : 0) : ((x * magic) >> 16);
}

// dW[m][n][S*u+r] += sum over the flattened (sample, time) columns.  Block tile: 128 rows
// (m) x 128 columns ((n,r),u = 128/U virtual channels x U taps), contraction chunks of TK
// columns, double buffered.  LO_ID / HI_ID: that operand has the identity transform (the
// gradient operand always has), so its staging is a plain copy.
// MB x NBT: the block tile (128 x 128, or 64 x 64 for the first layers whose M <= 64 rows and
// N*S <= 64/U virtual channels would leave 3/4 and more of the big tile empty).
template <int U, int TK, bool LO_ID, bool HI_ID, int MB, int NBT>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int S = 32 / U;
  constexpr int CVW = NBT / U;       // virtual channels per block (NBT output columns)
  constexpr int NI = MB / 64, NJ = NBT / 64;   // 32x32 MFMA blocks per wave (2 x 2 waves)
  constexpr int NN = CVW / S;        // real hi channels per block
  constexpr int AST = TK + 4;        // lo row stride: 16-B aligned rows, conflict-free b128 reads
  constexpr int NJ8 = TK / 8;        // groups of 8 contraction columns

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RLw = a.RLw;
  float* Al0 = smem;                  // [2][MB*AST]
  float* Bl0 = Al0 + 2 * MB * AST;    // [2][CVW*RLw]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  if (a.prio_mode == 1) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned hsh = (lin * 2654435761u) >> 30;
    if (hsh == 1) __builtin_amdgcn_s_setprio(1);
    else if (hsh == 2) __builtin_amdgcn_s_setprio(2);
    else if (hsh == 3) __builtin_amdgcn_s_setprio(3);
  }
  const int cv0 = blockIdx.x * CVW;
  const int m0 = blockIdx.y * MB;
  const int split_beg = blockIdx.z * a.cols_per_split;
  const int split_end = min(split_beg + a.cols_per_split, a.Ctot);
  if (split_beg >= split_end) return;
  const int nch = (split_end - split_beg + TK - 1) / TK;
  const int Ls = a.Ls;

  // ---- MFMA operand offsets.  Lane (row/col l31, half h) supplies contraction columns
  // k' = 8j + 4h + i (i = 0..3) of group j: one ds_read_b128 of the lo tile per row block.
  int aoff[NI], bbase[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) aoff[i] = (wm * (MB / 2) + 32 * i + l31) * AST + 4 * h;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cc = wn * (NBT / 2) + 32 * j + l31;
    bbase[j] = (cc / U) * RLw + cc % U;
  }

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // ---- lo staging: thread owns one float4 (4 consecutive columns; Ls % 4 == 0 keeps them
  // in one sample) of rows ar0 + RPA*i.  Row bases / transforms never change.
  constexpr int F4A = TK / 4;
  constexpr int RPA = 256 / F4A;
  constexpr int NPA = MB / RPA;
  const int kc4 = tid % F4A, ar0 = tid / F4A;
  const float* arow[NPA];
  bool arow_ok[NPA], arow_s1[NPA];
  ChanXf axf[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    int m = m0 + ar0 + RPA * i;
    arow_ok[i] = m < a.M;
    m = arow_ok[i] ? m : 0;
    arow_s1[i] = m >= a.lo.C0;
    arow[i] = arow_s1[i] ? a.lo.p1 + (size_t)(m - a.lo.C0) * Ls : a.lo.p0 + (size_t)m * Ls;
    if (!LO_ID) axf[i] = segan_chan_xf(a.lo, m);
  }
  // ---- hi staging: thread owns LDS position tid (< RLw <= 256) of all CVW channels
  const float* brow[NN];
  bool brow_s1[NN];
  ChanXf bxf[NN];
#pragma unroll
  for (int c = 0; c < NN; ++c) {
    int n = cv0 / S + c;
    n = n < a.N ? n : 0;
    brow_s1[c] = n >= a.hi.C0;
    brow[c] = brow_s1[c] ? a.hi.p1 + (size_t)(n - a.hi.C0) * a.Lhi : a.hi.p0 + (size_t)n * a.Lhi;
    if (!HI_ID) bxf[c] = segan_chan_xf(a.hi, n);
  }

  f32x4 areg[NPA];
  float breg[CVW];
  bool a_ok = false;
  unsigned b_ok = 0u;

  auto load_chunk = [&](int ch) {
    const int col0 = split_beg + ch * TK;
    const int b0 = col0 / Ls;
    const int t_first = col0 - b0 * Ls;
    // ---- lo ----
    {
      const int c4 = 4 * kc4;
      a_ok = col0 + c4 < split_end;
      const int x = t_first + c4;
      const int sd = wg_sdiv<TK>(x, Ls, a.ls_magic);
      int bb = b0 + sd;
      bb = (a_ok && bb < a.B) ? bb : 0;
      const int t = x - sd * Ls;
      const int o0 = bb * a.lo.C0 * Ls + t, o1 = bb * a.lo.C1 * Ls + t;
#pragma unroll
      for (int i = 0; i < NPA; ++i)
        areg[i] = *reinterpret_cast<const f32x4*>(arow[i] + (arow_s1[i] ? o1 : o0));
    }
    // ---- hi ----
    int s = 0, tau = 0;
    if (Ls >= TK) {
      const int len0 = min(Ls - t_first, TK);
      if (tid < len0 + a.H) { s = 0; tau = t_first + tid; }
      else { s = 1; tau = tid - (len0 + a.H); }
    } else {
      // chunks start on a sample boundary only when Ls divides TK; general decode otherwise
      const int len0 = min(Ls - t_first, TK);
      if (tid < len0 + a.H) { s = 0; tau = t_first + tid; }
      else {
        const int jj = tid - (len0 + a.H);
        const int q = (jj * a.per_magic) >> 16;
        s = 1 + q;
        tau = jj - q * (Ls + a.H);
      }
    }
    const int bs = b0 + s;
    const bool bok = tid < RLw && bs < a.B;
    const int bsc = bok ? bs : 0;
    const int so0 = bsc * a.hi.C0 * a.Lhi, so1 = bsc * a.hi.C1 * a.Lhi;
    int poff[S];
    b_ok = 0u;
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int idx = segan_hi_index(S * tau + r, a.Lhi, a.padL, a.mode, a.roll);
      poff[r] = (bok && idx >= 0) ? idx : 0;
      if (bok && idx >= 0) b_ok |= 1u << r;
    }
#pragma unroll
    for (int c = 0; c < CVW; ++c)
      breg[c] = brow[c / S][(brow_s1[c / S] ? so1 : so0) + poff[c % S]];
  };
  auto store_chunk = [&](int buf) {
    float* Al = Al0 + buf * (MB * AST);
    float* Bl = Bl0 + buf * (CVW * RLw);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const bool ok = a_ok && arow_ok[i];
      f32x4 v = areg[i];
      if (!LO_ID) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = segan_apply_xf(axf[i], v[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
      *reinterpret_cast<f32x4*>(Al + (ar0 + RPA * i) * AST + 4 * kc4) = v;
    }
    if (tid < RLw) {
#pragma unroll
      for (int c = 0; c < CVW; ++c) {
        const bool ok = (cv0 + c) < a.Cv && ((b_ok >> (c % S)) & 1u);
        float v = breg[c];
        if (!HI_ID) v = segan_apply_xf(bxf[c / S], v);
        Bl[c * RLw + tid] = ok ? v : 0.0f;
      }
    }
  };

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) load_chunk(ch + 1);
    const float* Al = Al0 + buf * (MB * AST);
    const float* Bl = Bl0 + buf * (CVW * RLw);
    // LDS position of contraction column k' = 8j + 4h (+i): sample s of the chunk sits s*H
    // further right; 4 | Ls keeps the 4 columns of a group in one sample.
    const int t_first = (split_beg + ch * TK) % Ls;
    int bpos[NJ8][NJ];
#pragma unroll
    for (int j = 0; j < NJ8; ++j) {
      const int k0 = 8 * j + 4 * h;
      const int p = k0 + wg_sdiv<TK>(t_first + k0, Ls, a.ls_magic) * a.H;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) bpos[j][jj] = bbase[jj] + p;
    }
    f32x4 af0[NI], af1[NI];
    float bv0[NJ], bv1[NJ];
    auto read_a = [&](int j, f32x4 (&af)[NI]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) af[i] = *reinterpret_cast<const f32x4*>(Al + aoff[i] + 8 * j);
    };
    auto read_b = [&](int s, float (&bv)[NJ]) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) bv[jj] = Bl[bpos[s / 4][jj] + (s & 3)];
    };
    auto mma = [&](const f32x4 (&af)[NI], int e, const float (&bv)[NJ]) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bv[jj], acc[i][jj], 0, 0, 0);
    };
    read_a(0, af0);
    read_b(0, bv0);
#define SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int j = 0; j < NJ8; j += 2) {
      // group j (af0), then group j+1 (af1); B one step ahead in alternating sets; the
      // sched_barriers pin "next reads, then this step's MFMAs"
      read_a(j + 1, af1);
      read_b(4 * j + 1, bv1); SB; mma(af0, 0, bv0); SB;
      read_b(4 * j + 2, bv0); SB; mma(af0, 1, bv1); SB;
      read_b(4 * j + 3, bv1); SB; mma(af0, 2, bv0); SB;
      read_b(4 * j + 4, bv0); SB; mma(af0, 3, bv1); SB;
      if (j + 2 < NJ8) read_a(j + 2, af0);
      read_b(4 * j + 5, bv1); SB; mma(af1, 0, bv0); SB;
      read_b(4 * j + 6, bv0); SB; mma(af1, 1, bv1); SB;
      read_b(4 * j + 7, bv1); SB; mma(af1, 2, bv0); SB;
      if (4 * j + 8 < TK / 2) read_b(4 * j + 8, bv0);
      SB; mma(af1, 3, bv1); SB;
    }
#undef SB
    if (ch + 1 < nch) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: dw[m][n][S*u + r] += acc ----
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cc = wn * (NBT / 2) + 32 * j + l31;
    const int cv = cv0 + cc / U;
    const int u = cc % U;
    const int n = cv / S, r = cv % S;
    const int k = S * u + r;
    if (cv >= a.Cv || k >= a.K) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (MB / 2) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < a.M) atomicAdd(a.dw + ((size_t)m * a.N + n) * a.K + k, acc[i][j][e]);
      }
  }
}

template <int U, bool LO_ID, bool HI_ID, int MB, int NBT>
static int launch_wgrad_tile(WgradArgs& a, hipStream_t st) {
  constexpr int CVW = NBT / U;
  constexpr int TK = 32;
  int NS;
  if (a.Ls >= TK) NS = (a.Ls % TK == 0) ? 1 : 2;
  else NS = (TK % a.Ls == 0) ? TK / a.Ls : (TK + a.Ls - 2) / a.Ls + 1;
  a.H = U - 1;
  a.RLw = TK + NS * a.H;
  // row stride = 8 (mod 32): the 4 channels x 8 taps a half-wave reads hit 32 distinct banks
  a.RLw += (8 - a.RLw % 32 + 32) % 32;
  if (a.RLw > 256 || a.Ls % 4 != 0) {
    segan_set_error("wgrad: low-rate length %d unsupported for stride %d (needs a multiple of 4, "
                    "and >= %d)", a.Ls, 32 / U, U / 2);
    return SEGAN_EUNSUPPORTED;
  }
  if ((long)a.B * a.M * a.Ls >= (1L << 31) || (long)a.B * a.N * a.Lhi >= (1L << 31)) {
    segan_set_error("wgrad: operand exceeds the 2^31 element indexing limit");
    return SEGAN_EUNSUPPORTED;
  }
  if (int e = segan_src_defaults(&a.lo, st, "wgrad(lo)")) return e;
  if (int e = segan_src_defaults(&a.hi, st, "wgrad(hi)")) return e;
  static const int prio_env = [] { const char* e = getenv("SEGAN_PRIO"); return e ? atoi(e) : 0; }();
  a.prio_mode = prio_env;
  a.ls_magic = (65536 + a.Ls - 1) / a.Ls;
  a.per_magic = (65536 + a.Ls + a.H - 1) / (a.Ls + a.H);
  const int ncol = ceil_div(a.Cv, CVW);
  const int nrow = ceil_div(a.M, MB);
  // split the (b,t) contraction so the grid has a few workgroups per CU
  const int tiles = ncol * nrow;
  const int chunks = ceil_div(a.Ctot, TK);
  static const int tgt_env = [] { const char* e = getenv("SEGAN_WGRAD_BLOCKS"); return e ? atoi(e) : 0; }();
  int nsplit = ceil_div(tgt_env > 0 ? tgt_env : 1536, tiles);
  if (nsplit > chunks / 4) nsplit = chunks / 4;   // at least 4 chunks per workgroup
  if (nsplit < 1) nsplit = 1;
  const int chunks_per = ceil_div(chunks, nsplit);
  nsplit = ceil_div(chunks, chunks_per);
  a.cols_per_split = chunks_per * TK;
  const size_t lds = (size_t)(2 * MB * (TK + 4) + 2 * CVW * a.RLw) * sizeof(float);
  auto kern = wgrad_kernel<U, TK, LO_ID, HI_ID, MB, NBT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(ncol, nrow, nsplit), dim3(256), lds, st, a);
  return segan_check_launch("wgrad_kernel");
}

template <int U, bool LO_ID, bool HI_ID>
static int launch_wgrad_x(WgradArgs& a, hipStream_t st) {
  static const bool small_on = [] { const char* e = getenv("SEGAN_WGRAD_SMALL"); return !e || atoi(e) != 0; }();
  // edge layers (1-2 channels on the hi side: N*S <= 64/U virtual channels): 64 columns
  // suffice, and 64 rows when M <= 64
  if (small_on && a.Cv <= 64 / U) {
    if (a.M <= 64) return launch_wgrad_tile<U, LO_ID, HI_ID, 64, 64>(a, st);
    return launch_wgrad_tile<U, LO_ID, HI_ID, 128, 64>(a, st);
  }
  return launch_wgrad_tile<U, LO_ID, HI_ID, 128, 128>(a, st);
}

template <int U>
static int launch_wgrad_t(WgradArgs& a, hipStream_t st) {
  const bool lo_id = !a.lo.scale && !a.lo.shift && !a.lo.slope;
  const bool hi_id = !a.hi.scale && !a.hi.shift && !a.hi.slope;
  if (lo_id && hi_id) return launch_wgrad_x<U, true, true>(a, st);
  if (lo_id) return launch_wgrad_x<U, true, false>(a, st);
  if (hi_id) return launch_wgrad_x<U, false, true>(a, st);
  return launch_wgrad_x<U, false, false>(a, st);
}

// ====================================================================================
// weight packing
// ====================================================================================
// Both packings are [*, K] -> [K', *] transposes of a 64 x 32 tile through LDS so that the
// global reads (31 contiguous taps per (m,n)) and the writes (64 contiguous m / n) are both
// coalesced.  grid.x = tiles of 64 along the transposed axis, grid.y = the other axis.
__global__ __launch_bounds__(256) void pack_f_kernel(const float* __restrict__ w,
                                                     float* __restrict__ wf, int M, int N, int K,
                                                     int S, int U, int pitch, int rows) {
  __shared__ float t[64][33];
  const int n = blockIdx.y;                 // may run past N into the zero padding rows
  const int m0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 32; e += 256) {
    const int ml = e >> 5, k = e & 31;
    const int m = m0 + ml;
    t[ml][k] = (m < M && n < N && k < K) ? w[((size_t)m * N + n) * K + k] : 0.0f;
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
  for (int e = tid; e < 64 * 32; e += 256) {
    const int nl = e >> 5, k = e & 31;
    const int n = n0 + nl;
    t[nl][k] = (m < M && n < N && k < K) ? w[((size_t)m * N + n) * K + k] : 0.0f;
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
static bool stride_ok(int S) { return S == 1 || S == 2 || S == 4; }

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
                       wf, M, N, K, S, U, pitch, rows);
  }
  if (wt) {
    const int NP = t_np(N, S);
    const int pitch = t_pitch(N, S), rows = t_rows(M, S);
    hipLaunchKernelGGL(pack_t_kernel, dim3(ceil_div(NP, 64), ceil_div(rows, U)), dim3(256), 0, st,
                       w, wt, M, N, K, S, U, NP, pad_t, pitch, rows);
  }
  return segan_check_launch("pack_weights");
}

static int check_src(const segan_src* s, int C, const char* what) {
  SEGAN_REQUIRE(s != nullptr && s->p0 != nullptr, "%s: source is NULL", what);
  SEGAN_REQUIRE(s->C0 > 0 && s->C1 >= 0 && s->C0 + s->C1 == C,
                "%s: channel segments %d+%d != %d", what, s->C0, s->C1, C);
  SEGAN_REQUIRE(s->C1 == 0 || s->p1 != nullptr, "%s: second segment pointer is NULL", what);
  return SEGAN_OK;
}

static bool precision_ok(int p) { return p == 0 || p == 1 || p == 3; }

extern "C" int segan_conv1d_fwd(const segan_src* x, const void* wf, const float* bias, float* out,
                                int B, int N, int M, int L, int K, int S, int padL, int mode,
                                int roll, int precision, void* stream) {
  SEGAN_REQUIRE(precision_ok(precision), "conv1d_fwd: precision must be 0, 1 or 3");
  SEGAN_REQUIRE(stride_ok(S), "conv1d_fwd: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "conv1d_fwd: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && L > 0, "conv1d_fwd: bad sizes");
  SEGAN_REQUIRE(L % S == 0, "conv1d_fwd: length %d not divisible by stride %d", L, S);
  SEGAN_REQUIRE(wf && out, "conv1d_fwd: NULL pointer");
  SEGAN_REQUIRE(mode == SEGAN_PAD_REFLECT || mode == SEGAN_PAD_ZERO, "conv1d_fwd: bad pad mode");
  SEGAN_REQUIRE(mode != SEGAN_PAD_REFLECT || (padL < L && K - 1 - padL < L),
                "conv1d_fwd: reflect padding %d needs length > pad (L=%d)", padL, L);
  SEGAN_REQUIRE(roll > -L && roll < L, "conv1d_fwd: |roll| must be < L");
  if (int e = check_src(x, N, "conv1d_fwd")) return e;
  const int U = 32 / S;
  CorrArgs a = {};
  a.in = *x;
  a.wp = (const float*)wf;
  a.out0 = out; a.out1 = nullptr; a.bias = bias; a.halo = nullptr;
  a.B = B; a.Cv = N * S; a.Ktot = N * 32; a.RP = f_pitch(M); a.Rvalid = M;
  a.Tcols = L / S; a.Ctot = B * a.Tcols;
  a.Lin = L; a.padL = padL; a.mode = mode; a.roll = roll;
  a.win_start = 0; a.H = U - 1;
  a.NP = 1; a.Nout = 0;
  a.OC0 = M; a.OC1 = 0; a.Lout = a.Tcols; a.act = SEGAN_ACT_NONE;
  a.out0_elems = (size_t)B * M * a.Tcols;
  if (precision) return segan_corr_bf_f(a, U, wf, precision, (hipStream_t)stream);
  static const bool fsmall_on = [] { const char* e = getenv("SEGAN_FSMALL"); return !e || atoi(e) != 0; }();
  if (N <= 2 && fsmall_on) return launch_fsmall(a, M, N, S, (hipStream_t)stream);
  return launch_corr<true, false>(a, U, (hipStream_t)stream);
}

extern "C" int segan_deconv1d_dgrad(const float* dy, const void* wf, float* dx0, float* dx1, int B,
                                    int M, int M0, int N, int Ls, int K, int S, int pad,
                                    int precision, void* stream) {
  SEGAN_REQUIRE(precision_ok(precision), "deconv1d_dgrad: precision must be 0, 1 or 3");
  SEGAN_REQUIRE(stride_ok(S), "deconv1d_dgrad: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "deconv1d_dgrad: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && Ls > 0, "deconv1d_dgrad: bad sizes");
  SEGAN_REQUIRE(M0 >= 0 && M0 <= M, "deconv1d_dgrad: split %d outside [0,%d]", M0, M);
  SEGAN_REQUIRE(dy && wf, "deconv1d_dgrad: NULL pointer");
  SEGAN_REQUIRE(dx0 || dx1, "deconv1d_dgrad: both destinations NULL");
  const int U = 32 / S;
  CorrArgs a = {};
  a.in.p0 = dy; a.in.p1 = nullptr; a.in.C0 = N; a.in.C1 = 0;
  a.in.scale = a.in.shift = a.in.slope = nullptr;
  a.wp = (const float*)wf;
  a.bias = nullptr; a.halo = nullptr;
  a.B = B; a.Cv = N * S; a.Ktot = N * 32; a.RP = f_pitch(M); a.Rvalid = M;
  a.Tcols = Ls; a.Ctot = B * Ls;
  a.Lin = S * Ls; a.padL = pad; a.mode = SEGAN_PAD_ZERO; a.roll = 0;
  a.win_start = 0; a.H = U - 1;
  a.NP = 1; a.Nout = 0;
  if (M0 == 0) { a.out0 = dx1; a.OC0 = M; a.out1 = nullptr; a.OC1 = 0; }
  else { a.out0 = dx0; a.OC0 = M0; a.out1 = dx1; a.OC1 = M - M0; }
  a.Lout = Ls; a.act = SEGAN_ACT_NONE;
  a.out0_elems = (size_t)B * a.OC0 * Ls;
  a.out1_elems = (size_t)B * a.OC1 * Ls;
  if (precision) return segan_corr_bf_f(a, U, wf, precision, (hipStream_t)stream);
  return launch_corr<true, false>(a, U, (hipStream_t)stream);
}

extern "C" int segan_deconv1d_fwd(const segan_src* x, const void* wt, const float* w,
                                  const float* bias, float* y, int B, int M, int N, int Ls, int K,
                                  int S, int pad, int act, int precision, void* stream) {
  SEGAN_REQUIRE(precision_ok(precision), "deconv1d_fwd: precision must be 0, 1 or 3");
  SEGAN_REQUIRE(stride_ok(S), "deconv1d_fwd: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "deconv1d_fwd: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && Ls > 0, "deconv1d_fwd: bad sizes");
  SEGAN_REQUIRE(y && (wt || (w && N <= 2)), "deconv1d_fwd: NULL pointer");
  SEGAN_REQUIRE(pad >= 0 && K - 2 * pad - S == (K & 1),
                "deconv1d_fwd: K=%d S=%d pad=%d does not give an output of S*Ls samples", K, S, pad);
  SEGAN_REQUIRE(act == SEGAN_ACT_NONE || act == SEGAN_ACT_TANH, "deconv1d_fwd: bad activation");
  if (int e = check_src(x, M, "deconv1d_fwd")) return e;
  const int U = 32 / S;
  CorrArgs a = {};
  a.in = *x;
  a.wp = (const float*)wt;
  a.out0 = y; a.out1 = nullptr; a.bias = bias; a.halo = nullptr;
  a.NP = t_np(N, S); a.Nout = N;
  a.B = B; a.Cv = M; a.Ktot = M * U; a.RP = t_pitch(N, S); a.Rvalid = S * a.NP;
  a.Tcols = Ls; a.Ctot = B * Ls;
  a.Lin = Ls; a.padL = 0; a.mode = SEGAN_PAD_ZERO; a.roll = 0;
  int cmin = 1 << 30, cmax = 0;
  for (int r = 0; r < S; ++r) {
    const int c = (r + pad) / S;
    cmin = c < cmin ? c : cmin;
    cmax = c > cmax ? c : cmax;
  }
  for (int r = 0; r < 4; ++r) a.rowshift[r] = r < S ? (r + pad) / S - cmin : 0;
  a.win_start = cmin - (U - 1);
  a.H = U - 1 + (cmax - cmin);
  a.OC0 = N; a.OC1 = 0; a.Lout = S * Ls; a.act = act;
  a.o_padL = 0; a.o_roll = 0; a.o_padR = 0;
  a.out0_elems = (size_t)B * N * S * Ls;
  if (w && N <= 2) return launch_tsmall(a, w, K, M, N, S, pad, (hipStream_t)stream);
  if (precision && act == SEGAN_ACT_NONE)
    return segan_corr_bf_t(a, U, wt, precision, (hipStream_t)stream);
  SEGAN_REQUIRE(precision == 0, "deconv1d_fwd: tanh epilogue only on the fp32 path");
  return launch_corr<false, true>(a, U, (hipStream_t)stream);
}

extern "C" int segan_conv1d_dgrad(const float* da, const void* wt, const float* w, float* dx,
                                  float* halo, int B, int N, int M, int L, int K, int S, int padL,
                                  int roll, int precision, void* stream) {
  SEGAN_REQUIRE(precision_ok(precision), "conv1d_dgrad: precision must be 0, 1 or 3");
  SEGAN_REQUIRE(stride_ok(S), "conv1d_dgrad: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "conv1d_dgrad: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && L > 0 && L % S == 0, "conv1d_dgrad: bad sizes");
  SEGAN_REQUIRE(da && dx && halo && (wt || (w && N <= 2)), "conv1d_dgrad: NULL pointer");
  SEGAN_REQUIRE(roll > -L && roll < L, "conv1d_dgrad: |roll| must be < L");
  const int padR = K - 1 - padL;
  SEGAN_REQUIRE(padL >= 0 && padR >= 0 && padL < L && padR < L, "conv1d_dgrad: bad padding");
  const int U = 32 / S;
  const int Ls = L / S;
  hipStream_t st = (hipStream_t)stream;
  CorrArgs a = {};
  a.in.p0 = da; a.in.p1 = nullptr; a.in.C0 = M; a.in.C1 = 0;
  a.in.scale = a.in.shift = a.in.slope = nullptr;
  a.wp = (const float*)wt;
  a.out0 = dx; a.out1 = nullptr; a.bias = nullptr; a.halo = halo;
  a.NP = t_np(N, S); a.Nout = N;
  a.B = B; a.Cv = M; a.Ktot = M * U; a.RP = t_pitch(N, S); a.Rvalid = S * a.NP;
  // padded coordinates P = S*q + r in [0, L + padL + padR)
  a.Tcols = (L + padL + padR - 1) / S + 1;
  a.Ctot = B * a.Tcols;
  a.Lin = Ls; a.padL = 0; a.mode = SEGAN_PAD_ZERO; a.roll = 0;
  for (int r = 0; r < 4; ++r) a.rowshift[r] = 0;
  a.win_start = -(U - 1);
  a.H = U - 1;
  a.OC0 = N; a.OC1 = 0; a.Lout = L; a.act = SEGAN_ACT_NONE;
  a.o_padL = padL; a.o_roll = roll; a.o_padR = padR;
  a.out0_elems = (size_t)B * N * L;
  a.halo_elems = (size_t)B * N * (padL + padR);
  int e = (w && N <= 2) ? launch_tsmall(a, w, K, M, N, S, 0, st)
          : precision   ? segan_corr_bf_t(a, U, wt, precision, st)
                        : launch_corr<false, true>(a, U, st);
  if (e) return e;
  if (padL + padR > 0) {
    const int rows = B * N;
    // left targets are 1..padL, right targets L-1-padR..L-2: disjoint iff padL < L-1-padR
    const int per_sample = (padL < L - 1 - padR) ? 1 : 0;
    const long nthreads = per_sample ? (long)rows * (padL + padR) : rows;
    hipLaunchKernelGGL(fold_halo_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                       dx, halo, rows, L, padL, padR, roll, per_sample);
    return segan_check_launch("fold_halo_kernel");
  }
  return SEGAN_OK;
}

extern "C" size_t segan_wgrad_scratch_bytes(int B, int M, int Ls, int precision) {
  if (precision == SEGAN_PREC_FP32 || B <= 0 || M <= 0 || Ls <= 0) return 0;
  return segan_wgrad_bf_scratch_bytes(B, M, Ls, precision == SEGAN_PREC_BF16 ? 1 : 3);
}

extern "C" int segan_wgrad(const segan_src* lo, const segan_src* hi, float* dw, int B, int M, int N,
                           int Ls, int K, int S, int padL, int mode, int roll, int precision,
                           void* scratch, void* stream) {
  SEGAN_REQUIRE(precision_ok(precision), "wgrad: bad precision %d", precision);
  SEGAN_REQUIRE(stride_ok(S), "wgrad: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "wgrad: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && Ls > 0, "wgrad: bad sizes");
  SEGAN_REQUIRE(dw != nullptr, "wgrad: dw is NULL");
  SEGAN_REQUIRE(mode == SEGAN_PAD_REFLECT || mode == SEGAN_PAD_ZERO, "wgrad: bad pad mode");
  if (int e = check_src(lo, M, "wgrad(lo)")) return e;
  if (int e = check_src(hi, N, "wgrad(hi)")) return e;
  const int L = S * Ls;
  SEGAN_REQUIRE(roll > -L && roll < L, "wgrad: |roll| must be < L");
  WgradArgs a = {};
  a.lo = *lo; a.hi = *hi; a.dw = dw;
  a.B = B; a.M = M; a.N = N; a.K = K; a.Ls = Ls; a.Lhi = L;
  a.Cv = N * S; a.padL = padL; a.mode = mode; a.roll = roll;
  a.Ctot = B * Ls;
  hipStream_t st = (hipStream_t)stream;
  if (precision != SEGAN_PREC_FP32) {
    a.lo_pk = scratch;
    return segan_wgrad_bf(a, 32 / S, precision == SEGAN_PREC_BF16 ? 1 : 3, st);
  }
  switch (S) {
    case 4: return launch_wgrad_t<8>(a, st);
    case 2: return launch_wgrad_t<16>(a, st);
    default: return launch_wgrad_t<32>(a, st);
  }
}
