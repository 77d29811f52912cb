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
// C ABI: forward and data-gradient entry points
// ====================================================================================
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
  if (N <= 2 && fsmall_on) return segan_launch_fsmall(a, M, N, S, (hipStream_t)stream);
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
  if (w && N <= 2) return segan_launch_tsmall(a, w, K, M, N, S, pad, (hipStream_t)stream);
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
  int e = (w && N <= 2) ? segan_launch_tsmall(a, w, K, M, N, S, 0, st)
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
