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

// Tile geometry: MB rows x NB columns per 256-thread workgroup, waves WM x (4/WM).
//   F form: rows = output channels m; waves 2x2.
//   T form: rows = (phase r, channel n) with ALL S phases of MB/S channels in one tile and
//           waves 1x4 (NB=128) so that one lane ends up holding the S consecutive output
//           samples S*q..S*q+S-1 of a channel: full-line stores instead of stride-S ones.
//
// What the instruction stream beside the MFMAs costs was measured (profiles/r02_mfma_issue_cost
// .json): v_mfma_f32_32x32x2_f32 issues every 64 cycles per SIMD, and every VALU instruction
// any wave of that SIMD issues takes ~5 of those cycles, every VMEM instruction ~25, a
// ds_write_b32 ~4, a ds_write_b128 ~12; SALU and ds_read_b32 are (almost) free.  The staging
// code of corr2_kernel is therefore built from buffer loads whose per-lane offsets never
// change (wave-uniform parts go through the SGPR offset, masking through the descriptor's
// range check), LDS-DMA for the weight tile, and per-wave-uniform transforms.
template <int MB, int NB, int WM, int U>
struct TileGeom {
  static constexpr int S = 32 / U;
  static constexpr int WN = 4 / WM;
  static constexpr int NI = MB / (32 * WM);
  static constexpr int NJ = NB / (32 * WN);
  static constexpr int NPT = MB / S;   // T form: channels per tile
};

// which tile a workgroup works on, and which chunk range of it (stream-K pieces)
struct TileWork {
  int tile, c0, c1;
  bool partial;
  int piece;   // slab index of a partial piece inside the workgroup's two slabs
};

// Work decomposition (data-parallel + stream-K hybrid).  The first a.sk_nfull tiles are
// whole-tile work items, strided over the grid.  The remaining tiles — fewer than the grid,
// i.e. the partial last round that would leave CUs idle — are cut in the (tile, chunk)
// iteration space into equal contiguous ranges, one per workgroup.  A piece of a cut tile is
// written as an accumulator slab to the stream-K workspace; corr_fixup_kernel adds the slabs
// of a tile in chunk order (a FIXED order: results are bit-reproducible) and stores it.
struct TileIter {
  int tileA;
  long unit, unit_end;
  int first_tile;
  __device__ __forceinline__ void init(const CorrArgs& a, int nch) {
    tileA = blockIdx.x;
    unit = (long)blockIdx.x * a.sk_units;
    unit_end = min(unit + (long)a.sk_units, a.sk_total);
    first_tile = (int)(unit / nch);
  }
  __device__ __forceinline__ bool next(const CorrArgs& a, int nch, TileWork& w) {
    if (tileA < a.sk_nfull) {
      // whole tiles, one round of gridDim.x (a multiple of 8, or the whole launch) at a time,
      // each XCD on a contiguous range of the round
      const int rd = tileA / (int)gridDim.x;
      const int left = a.sk_nfull - rd * (int)gridDim.x;
      w.tile = rd * (int)gridDim.x + xcd_remap(blockIdx.x, left < (int)gridDim.x ? left : (int)gridDim.x);
      w.c0 = 0; w.c1 = nch; w.partial = false; w.piece = 0;
      tileA += gridDim.x;
      return true;
    }
    if (unit < unit_end) {
      const int t = (int)(unit / nch);
      w.c0 = (int)(unit - (long)t * nch);
      w.c1 = min(nch, w.c0 + (int)(unit_end - unit));
      unit += w.c1 - w.c0;
      w.tile = a.sk_nfull + t;
      w.partial = (w.c0 != 0) || (w.c1 != nch);
      w.piece = t - first_tile;
      return true;
    }
    return false;
  }
};

// per-lane column bookkeeping of a tile: which (sample, time) the lane's NJ columns are and
// where they sit in the LDS activation tile
template <int NB, int WN, int NJ>
__device__ __forceinline__ void lane_columns(const CorrArgs& a, const ColTile& ct, int wn, int l31,
                                             int h, int (&col_b)[NJ], int (&col_t)[NJ],
                                             int (&boff)[NJ]) {
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
}

// accumulator slab of a stream-K piece: [NI*NJ*4][256] float4, thread-minor (coalesced 16-byte
// accesses)
template <int NI, int NJ>
__device__ __forceinline__ void slab_store(float* slab, const f32x16 (&acc)[NI][NJ], int tid) {
  f32x4* s4 = reinterpret_cast<f32x4*>(slab);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                         acc[i][j][4 * q + 3]};
        s4[((i * NJ + j) * 4 + q) * 256 + tid] = v;
      }
}
template <int NI, int NJ>
__device__ __forceinline__ void slab_add(const float* slab, f32x16 (&acc)[NI][NJ], int tid) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = s4[((i * NJ + j) * 4 + q) * 256 + tid];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
      }
}

// ---- epilogue: bias, optional tanh, store of a finished tile ----
template <int MB, int NB, int WM, int U, bool OUT_HI>
__device__ __forceinline__ void corr_store_tile(
    const CorrArgs& a, const f32x16 (&acc)[TileGeom<MB, NB, WM, U>::NI][TileGeom<MB, NB, WM, U>::NJ],
    int m0, int n0, int wm, int h, const int (&col_b)[TileGeom<MB, NB, WM, U>::NJ],
    const int (&col_t)[TileGeom<MB, NB, WM, U>::NJ]) {
  using G = TileGeom<MB, NB, WM, U>;
  constexpr int S = G::S, NI = G::NI, NJ = G::NJ, NPT = G::NPT;
  if (!OUT_HI) {
    // Interior tiles — all rows valid and in one destination, plain stores — take the fast form:
    // buffer stores whose per-lane byte offsets are computed once per column (masked columns
    // carry an out-of-range offset: the store is dropped), the row inside the tile goes through
    // the scalar offset: no address arithmetic, row test or destination select per element (the
    // generic form below spends ~30 VALU per stored value, a tenth of a 64-chunk tile's time).
    {
      float* fdst = m0 < a.OC0 ? a.out0 : a.out1;
      const int foc = m0 < a.OC0 ? a.OC0 : a.OC1;
      const int foch = m0 < a.OC0 ? m0 : m0 - a.OC0;
      const long fbytes = (long)a.B * foc * a.Lout * 4;
      if (a.act == SEGAN_ACT_NONE && m0 + MB <= a.Rvalid && (m0 + MB <= a.OC0 || m0 >= a.OC0) &&
          fdst != nullptr && fbytes < 0x7fffffffL) {
        const __amdgpu_buffer_rsrc_t ors =
            __builtin_amdgcn_make_buffer_rsrc(fdst, 0, (int)fbytes, 0x00020000);
        int ovo[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          ovo[j] = col_b[j] < 0 ? (int)0x80000000u
                                : ((col_b[j] * foc + foch + 32 * wm * NI + 4 * h) * a.Lout + col_t[j]) * 4;
        const int rowstep = a.Lout * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rl = 32 * i + (e & 3) + 8 * (e >> 2);
            float bs = 0.0f;
            if (a.bias) bs = epi_bias(a.bias, m0 + 32 * wm * NI, rl, h);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][e] + bs), ors,
                                                    ovo[j], rl * rowstep, 0);
          }
        }
        return;
      }
    }
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
        const float bs = a.bias ? a.bias[row] : 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (col_b[j] < 0) continue;
          float v = acc[i][j][e] + bs;
          if (a.act == SEGAN_ACT_TANH) v = tanhf(v);
          dst[((size_t)col_b[j] * oc + och) * (size_t)a.Lout + col_t[j]] = v;
        }
      }
    }
  } else {
    // HI store.  Row block ib of the tile is phase r = 32*ib / NPT of channels n0 + nl.
    constexpr bool QUAD = (S == 4 && WM == 1 && NI == 4);  // lane holds all 4 phases of (n, q)
    const int hl = a.o_padL + a.o_padR;
    const long obytes = (long)a.B * a.Nout * a.Lout * 4;
    const bool qfast = obytes < 0x7fffffffL;      // 32-bit byte offsets reach every output element
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        a.out0, 0, (int)(qfast ? obytes : 0), 0x00020000);
    if (QUAD && qfast && a.act == SEGAN_ACT_NONE && a.o_padL == 0 && a.o_roll == 0 &&
        a.halo == nullptr && n0 + NPT <= a.Nout && a.Lout == 4 * a.Tcols) {
      // deconv forward: a lane's four phase accumulators are four consecutive output samples
      int ovo[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        ovo[j] = col_b[j] < 0 ? (int)0x80000000u
                              : ((col_b[j] * a.Nout + n0 + 4 * h) * a.Lout + 4 * col_t[j]) * 4;
      const int rowstep = a.Lout * 4;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nl = (e & 3) + 8 * (e >> 2);
        float bs = 0.0f;
        if (a.bias) bs = epi_bias(a.bias, n0, nl, h);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
          const u32x4s o = {__builtin_bit_cast(unsigned, acc[0 % NI][j][e] + bs),
                            __builtin_bit_cast(unsigned, acc[1 % NI][j][e] + bs),
                            __builtin_bit_cast(unsigned, acc[2 % NI][j][e] + bs),
                            __builtin_bit_cast(unsigned, acc[3 % NI][j][e] + bs)};
          __builtin_amdgcn_raw_buffer_store_b128(o, qrs, ovo[j] + nl * rowstep, 0, 0);   // (*)
        }
      }
      return;
    }
    int cb[NJ];          // columns still to be stored by the generic form below (-1: done / masked)
#pragma unroll
    for (int j = 0; j < NJ; ++j) cb[j] = col_b[j];
    if (QUAD && qfast && a.act == SEGAN_ACT_NONE && n0 + NPT <= a.Nout) {
      // conv data gradient (HI store with a left pad, a halo and possibly a roll): everything that
      // depends on the COLUMN only — sample, the first of the lane's four output positions, the
      // roll's wrap — is worked out once per column instead of once per stored vector; interior
      // columns (all but the ~8 at a row's ends and the one on the wrap point) then store one
      // 16-byte vector per channel row (the row offset added to the vector offset: (*) in segan_conv_shared.h).  The generic form below
      // (~30 VALU per stored vector) costs a bf16 tile 40 % of its time and an fp32 tile 5-10 %.
      int ovo[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ovo[j] = (int)0x80000000u;
        if (col_b[j] >= 0) {
          const int i0 = 4 * col_t[j] - a.o_padL;
          int ib = i0 - a.o_roll;
          if (ib < 0) ib += a.Lout;
          if (ib >= a.Lout) ib -= a.Lout;
          if (i0 >= 0 && i0 + 3 < a.Lout && ib + 3 < a.Lout) {
            ovo[j] = ((col_b[j] * a.Nout + n0 + 4 * h) * a.Lout + ib) * 4;
            cb[j] = -1;
          }
        }
      }
      const int rowstep = a.Lout * 4;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nl = (e & 3) + 8 * (e >> 2);
        float bs = 0.0f;
        if (a.bias) bs = epi_bias(a.bias, n0, nl, h);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
          const u32x4s o = {__builtin_bit_cast(unsigned, acc[0 % NI][j][e] + bs),
                            __builtin_bit_cast(unsigned, acc[1 % NI][j][e] + bs),
                            __builtin_bit_cast(unsigned, acc[2 % NI][j][e] + bs),
                            __builtin_bit_cast(unsigned, acc[3 % NI][j][e] + bs)};
          __builtin_amdgcn_raw_buffer_store_b128(o, qrs, ovo[j] + nl * rowstep, 0, 0);   // (*)
        }
      }
      bool left = false;
#pragma unroll
      for (int j = 0; j < NJ; ++j) left = left || cb[j] >= 0;
      if (!left) return;
    }
    // Stride 2 (round 6): a lane holds BOTH phases of its channels — in row blocks ib and ib + NI/2
    // (64 / 32 channels per tile), or in elements e and e + 8 of the one block of a 16-channel tile —
    // i.e. two consecutive output samples: one 8-byte store per (channel, column) for the deconv
    // forward and the interior columns of the conv data gradient, offsets once per column as in the
    // stride-4 form above.  The stride-2 T form used to take the generic per-element path below
    // (~30 VALU per stored value).
    constexpr bool PAIR = (S == 2 && WM == 1 && (NI == 4 || NI == 2 || NI == 1));
    if (PAIR && qfast && a.act == SEGAN_ACT_NONE && n0 + NPT <= a.Nout) {
      int ovo[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ovo[j] = (int)0x80000000u;
        if (col_b[j] >= 0) {
          const int i0 = 2 * col_t[j] - a.o_padL;
          int ib = i0 - a.o_roll;
          if (ib < 0) ib += a.Lout;
          if (ib >= a.Lout) ib -= a.Lout;
          if (i0 >= 0 && i0 + 1 < a.Lout && ib + 1 < a.Lout) {
            ovo[j] = ((col_b[j] * a.Nout + n0 + 4 * h) * a.Lout + ib) * 4;
            cb[j] = -1;
          }
        }
      }
      const int rowstep = a.Lout * 4;
      constexpr int NBLK = NI >= 2 ? NI / 2 : 1;      // 32-channel blocks of the tile
      constexpr int NE = NI >= 2 ? 16 : 8;            // channel slots per block and lane
#pragma unroll
      for (int bk = 0; bk < NBLK; ++bk) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int nl = 32 * bk + (e & 3) + 8 * (e >> 2);
          float bs = 0.0f;
          if (a.bias) bs = epi_bias(a.bias, n0, nl, h);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
            const float p0 = acc[bk][j][e];
            const float p1 = NI >= 2 ? acc[(bk + NI / 2) % NI][j][e] : acc[0][j][(e + 8) % 16];
            const u32x2s o = {__builtin_bit_cast(unsigned, p0 + bs), __builtin_bit_cast(unsigned, p1 + bs)};
            __builtin_amdgcn_raw_buffer_store_b64(o, qrs, ovo[j] + nl * rowstep, 0, 0);   // (*)
          }
        }
      }
      bool left = false;
#pragma unroll
      for (int j = 0; j < NJ; ++j) left = left || cb[j] >= 0;
      if (!left) return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (cb[j] < 0) continue;
        const int q = col_t[j];
        if (QUAD) {
          const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (n >= a.Nout) continue;
          const float bs = a.bias ? a.bias[n] : 0.0f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[r][j][e] + bs;
            if (a.act == SEGAN_ACT_TANH) v[r] = tanhf(v[r]);
          }
          const size_t rowoff = (size_t)col_b[j] * a.Nout + n;
          const int i0 = 4 * q - a.o_padL;
          if (qfast && i0 >= 0 && i0 + 3 < a.Lout) {
            // interior of the row (all but ~8 of its positions): the four phases are four
            // consecutive samples, also after the roll unless they straddle its wrap point
            int ib = i0 - a.o_roll;
            if (ib < 0) ib += a.Lout;
            if (ib >= a.Lout) ib -= a.Lout;
            if (ib + 3 < a.Lout) {
              typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
              const u32x4s o = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
              __builtin_amdgcn_raw_buffer_store_b128(o, qrs, (int)((rowoff * a.Lout + ib) * 4), 0, 0);
              continue;
            }
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
              a.out0[rowoff * (size_t)a.Lout + ii] = v[r];
            } else if (a.halo != nullptr) {
              if (ii < 0) a.halo[rowoff * hl + P] = v[r];
              else if (ii - a.Lout < a.o_padR) a.halo[rowoff * hl + a.o_padL + (ii - a.Lout)] = v[r];
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int rloc = 32 * (wm * NI + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int r = rloc / NPT;
            const int n = n0 + rloc % NPT;
            if (n >= a.Nout) continue;
            float v = acc[i][j][e] + (a.bias ? a.bias[n] : 0.0f);
            if (a.act == SEGAN_ACT_TANH) v = tanhf(v);
            const int P = S * q + r;
            int ii = P - a.o_padL;
            const size_t rowoff = (size_t)col_b[j] * a.Nout + n;
            if (ii >= 0 && ii < a.Lout) {
              if (a.o_roll != 0) {
                ii -= a.o_roll;
                if (ii < 0) ii += a.Lout;
                if (ii >= a.Lout) ii -= a.Lout;
              }
              a.out0[rowoff * (size_t)a.Lout + ii] = v;
            } else if (a.halo != nullptr) {
              if (ii < 0) a.halo[rowoff * hl + P] = v;
              else if (ii - a.Lout < a.o_padR) a.halo[rowoff * hl + a.o_padL + (ii - a.Lout)] = v;
            }
          }
        }
      }
    }
  }
}

// F form: rows of a tile that all go to a NULL destination (the z half of the first decoder
// layer's data gradient) are skipped
template <int MB, bool OUT_HI>
__device__ __forceinline__ bool tile_is_dead(const CorrArgs& a, int m0) {
  if (OUT_HI) return false;
  if (a.out0 == nullptr && m0 + MB <= a.OC0) return true;
  if (a.out1 == nullptr && m0 >= a.OC0) return true;
  return false;
}

// The MFMA loop over one staged chunk, shared by both kernels.  Operands of step s+2 are read
// from LDS before the MFMAs of step s are issued (three named register sets; everything is
// unrolled so all indices are static); sched_barrier pins that order so the LDS latency of the
// next operands is covered by the MFMAs instead of being exposed.
//
// K = 31 taps are padded to 32, so one contraction row in 32 multiplies zeros — half of one
// v_mfma_f32_32x32x2_f32 (two rows per instruction).  Two such halves are merged into one
// instruction (round 5; 1/32 of the F / T forms' matrix work):
//  * FMODE (F form, chunk = the 32 (r,u) rows of ONE input channel, the zero tap is row 31, paired
//    with row 30 in the last step).  Channels are processed in pairs (A = even, B = A + 1): the
//    packing stores A's row-30 weights in B's row 31 (segan_pack.hip, `f_pair`).  FMODE 1 = chunk
//    A with its partner following in this piece: steps 0..14 only, the row-30 activations are kept
//    in `b30` (all lanes read the lower half's address).  FMODE 2 = chunk B: 16 steps, the upper
//    half of the last step's activation operand is b30 — zero when the piece began with B (A's
//    row 30 is then added by the piece that held A: an A that ends its piece runs FMODE 0, whose
//    row 31 holds zeros).  FMODE 0 = the plain 16 steps.
//  * ZP >= 0 (T form, chunk = CV input channels x U taps, rows (c, u'); the zero tap is u' = 0 of
//    output phase ZP, i.e. of the 32-row blocks i with (32 i) / NPT == ZP).  Those blocks skip the
//    CV steps (c; u' = 0, 1) and run CV / 2 merged steps instead whose lower / upper half is row
//    (2p, 1) / (2p + 1, 1): operand offsets aoffx = aoff + h (U-1) MB, boffx = boff + h (RLs-1).
template <int MB, int U, int KC, int NI, int NJ, bool SHIFT, int FMODE = 0, int ZP = -1, int NPT = 32>
__device__ __forceinline__ void corr_mma_chunk(const float* Wl, const float* Il, int RLs,
                                               const int (&aoff)[NI], const int (&boff)[NJ],
                                               const int (&rsh)[NI], f32x16 (&acc)[NI][NJ],
                                               float (&b30)[NJ], int h, const int (&aoffx)[NI],
                                               const int (&boffx)[NJ]) {
  static_assert(FMODE == 0 || (!SHIFT && ZP < 0), "the pair trick of the F form has no row shifts");
  constexpr int NBI = SHIFT ? NI : 1;
  constexpr int CVC = KC / U;                           // channels (T) / phases (F) per chunk
  constexpr bool ZT = ZP >= 0 && CVC >= 2;
  // volatile LDS pointers: every operand read stays a ds_read_b32 with an immediate offset
  // (merged ds_read2 forms need a VALU address add per step, which costs MFMA issue time)
  typedef const volatile __attribute__((address_space(3))) float* ldsp;
  auto special = [](int i) { return ZT && (32 * i) / NPT == ZP; };
  auto skipped = [&](int i, int s) { return special(i) && ((2 * s) % U) == 0; };
  float av0[NI], av1[NI], bv0[NBI][NJ], bv1[NBI][NJ];
  auto read_step = [&](int s, float (&av)[NI], float (&bv)[NBI][NJ]) {
    const int kk = 2 * s;
    const int c = kk / U, u = kk % U;
    ldsp wr = (ldsp)(Wl + kk * MB);
    ldsp ir = (ldsp)(Il + c * RLs + u);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (!skipped(i, s)) av[i] = wr[aoff[i]];
#pragma unroll
    for (int i = 0; i < NBI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (!(SHIFT && skipped(i, s))) bv[i][j] = ir[boff[j] + (SHIFT ? rsh[i] : 0)];
  };
  constexpr int NS = FMODE == 1 ? KC / 2 - 1 : KC / 2;
  auto mma_step = [&](int s, const float (&av)[NI], float (&bv)[NBI][NJ]) {
    if (FMODE == 2 && s == NS - 1) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[0][j] = h ? b30[j] : bv[0][j];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (!skipped(i, s))
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[SHIFT ? i : 0][j], acc[i][j],
                                                           0, 0, 0);
  };
  if (FMODE == 1) {
    // row 30 = (phase S-1, tap U-2) of this channel, at the LOWER half's address in every lane
    ldsp ir = (ldsp)(Il + (CVC - 1) * RLs + (U - 2));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b30[j] = ir[boff[j] - h];
  }
  // the merged steps of the T form's zero-tap phase: operands first, MFMAs after the pipeline's
  // first two reads are in flight
  float avx[ZT ? CVC / 2 : 1][NI], bvx[ZT ? CVC / 2 : 1][NBI][NJ];
  if (ZT) {
#pragma unroll
    for (int p = 0; p < CVC / 2; ++p) {
      ldsp wr = (ldsp)(Wl + (2 * p * U + 1) * MB);
      ldsp ir = (ldsp)(Il + 2 * p * RLs + 1);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (special(i)) avx[p][i] = wr[aoffx[i]];
#pragma unroll
      for (int i = 0; i < NBI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (!SHIFT || special(i)) bvx[p][i][j] = ir[boffx[j] + (SHIFT ? rsh[i] : 0)];
    }
  }
  // operands are read TWO steps ahead of the MFMAs that use them (three register sets)
  float av2[NI], bv2[NBI][NJ];
  read_step(0, av0, bv0);
  read_step(1, av1, bv1);
  if (ZT) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < CVC / 2; ++p)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (special(i))
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(avx[p][i], bvx[p][SHIFT ? i : 0][j],
                                                             acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < NS; s += 3) {
    if (s + 2 < NS) read_step(s + 2, av2, bv2);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(s, av0, bv0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 3 < NS) read_step(s + 3, av0, bv0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < NS) mma_step(s + 1, av1, bv1);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 4 < NS) read_step(s + 4, av1, bv1);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < NS) mma_step(s + 2, av2, bv2);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ====================================================================================
// corr_kernel: the general form (any channel split, any window length up to 512, any pad
// mode with any transform).  Staging discipline: load_chunk() only ISSUES global loads —
// every address is clamped to a valid element, so there is no branch and no wait between
// them and they stay in flight under the MFMA loop; masking, the segan_src transform and the
// LDS writes happen in store_chunk(), after the compute of the previous chunk.
// ====================================================================================
template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, bool SHIFT, int MAXPOS, int KC>
__global__ __launch_bounds__(256, 2) void corr_kernel(const CorrArgs a) {
  using G = TileGeom<MB, NB, WM, U>;
  constexpr int S = G::S, WN = G::WN, NI = G::NI, NJ = G::NJ, NPT = G::NPT;
  constexpr int SI = IN_HI ? S : 1;   // indices per staged position
  constexpr int CV = KC / U;          // virtual channels per chunk
  static_assert(!OUT_HI || NPT % 32 == 0, "T-form tiles hold whole 32-row phase blocks");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RLs = a.RLs;
  float* Wl0 = smem;                  // [2][KC*MB]
  float* Il0 = smem + 2 * KC * MB;   // [2][CV*RLs]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int nch = (a.Ktot + KC - 1) / KC;
  TileIter it;
  it.init(a, nch);
  TileWork tw;
  while (it.next(a, nch, tw)) {
  const int c0 = tw.c0, c1 = tw.c1;
  const int rowtile = a.rt0 + tw.tile / a.ncoltiles;
  const int coltile = tw.tile % a.ncoltiles;
  const int m0 = rowtile * MB;          // F form: first row; T form: n0 = rowtile * NPT
  const int n0 = rowtile * NPT;
  if (tile_is_dead<MB, OUT_HI>(a, m0)) continue;
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
  lane_columns<NB, WN, NJ>(a, ct, wn, l31, h, col_b, col_t, boff);

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
    for (int p = 0; p < NPASS; ++p) {
      // f_pair packings keep the even partner's row-30 weights in row 31 of an odd channel (for
      // corr2_kernel's merged step): this kernel runs the plain 16 steps, where that row is the
      // zero tap
      if (IN_HI && a.f_pair && ((ch * KC + wrow + RPP * p) & 63) == 63)
        wreg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(Wl + (wrow + RPP * p) * MB + 4 * wc4) = wreg[p];
    }
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

  float b30_unused[NJ] = {};
  load_chunk(c0);
  store_chunk(c0, 0);
  __syncthreads();
  for (int ch = c0; ch < c1; ++ch) {
    const int buf = (ch - c0) & 1;
    if (ch + 1 < c1) load_chunk(ch + 1);
    corr_mma_chunk<MB, U, KC, NI, NJ, SHIFT>(Wl0 + buf * (KC * MB), Il0 + buf * (CV * RLs), RLs,
                                             aoff, boff, rsh, acc, b30_unused, h, aoff, boff);
    if (ch + 1 < c1) store_chunk(ch + 1, buf ^ 1);
    __syncthreads();
  }

  if (tw.partial)
    slab_store<NI, NJ>(a.sk_ws + (size_t)(blockIdx.x * 2 + tw.piece) * (MB * NB), acc, tid);
  else
    corr_store_tile<MB, NB, WM, U, OUT_HI>(a, acc, m0, n0, wm, h, col_b, col_t);
  }  // tile loop
}

// ====================================================================================
// corr2_kernel: the same contraction with a staging path that costs the matrix pipe almost
// nothing (see the head of this file).  Per chunk of KC = 32 contraction rows a wave issues
//   * NPASS buffer_load_dwordx4 ... lds : the weight tile goes HBM/L2 -> LDS directly;
//   * <= 4 buffer_load_dword with per-lane offsets that never change: a wave stages ONE
//     (virtual) channel of the chunk, positions lane + 64*i; everything that varies per chunk
//     (channel row, channel segment) is wave-uniform and goes through the scalar offset, and
//     masked positions carry an out-of-range offset (the load returns 0);
//   * the segan_src transform with wave-uniform scale / shift / slope (skipped for identity),
//     one ds_write_b32 per element.
// Preconditions (checked by the launcher, else corr_kernel runs): window of at most 256
// positions; the channel split C0 a multiple of the chunk's channels (T form); no shift with
// zero padding (a padded zero must stay zero after the transform).
// ====================================================================================
#define CORR2_KC 32
#define CORR2_NLD 4
template <int V>
struct IntC {
  static constexpr int value = V;
};

__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, bool SHIFT, bool XF, bool BLK = false>
// F form: 4 waves per SIMD (<= 128 VGPRs; 4 workgroups of 40 KB LDS per CU) as before round 5 — the
// pair loop needs 129-131 registers left alone; the T form (158-161) and the blocked-accumulation
// variants (2 waves) keep the bound of two
__global__ __launch_bounds__(256, (IN_HI && !BLK) ? 4 : 2) void corr2_kernel(const CorrArgs a) {
  using G = TileGeom<MB, NB, WM, U>;
  constexpr int S = G::S, WN = G::WN, NI = G::NI, NJ = G::NJ, NPT = G::NPT;
  constexpr int KC = CORR2_KC;
  constexpr int CV = KC / U;          // virtual channels per chunk (== S)
  constexpr int SL = 4 / CV;          // waves that share a channel, each a slice of positions
  constexpr int NLD = CORR2_NLD;
  static_assert(CV == S, "one real channel per chunk of a HI input");
  // a row shift is a property of a whole 32-row MFMA block (it moves the block's activation
  // operand): shifted T tiles hold whole 32-row phase blocks; unshifted ones may mix the phases of
  // NPT = 16 channels inside a block (the row decode of the weight tile and of the epilogue is per
  // element)
  static_assert(!OUT_HI || NPT % 32 == 0 || (!SHIFT && NPT % 4 == 0 && U != 8),
                "T-form tiles hold whole 32-row phase blocks");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RLs = a.RLs;              // padded row: a multiple of 64*SL, every lane may write
  float* Wl0 = smem;                  // [2][KC*MB]
  float* Il0 = smem + 2 * KC * MB;   // [2][CV*RLs]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int wc = wave % CV;           // the chunk channel this wave stages
  const int wslice = wave / CV;       // its slice of positions
  const int nld = a.nld;

  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, 0x7fffffff, 0x00020000);
  // weight tile: one LDS-DMA instruction moves 64 lanes x 16 B = 256/MB rows of MB floats
  constexpr int F4R = MB / 4;         // lanes per row
  constexpr int RPI = 64 / F4R;       // rows per instruction
  constexpr int NINS = KC / RPI;      // instructions per chunk and workgroup
  constexpr int NPASS = NINS / 4;     // per wave
  static_assert(NINS % 4 == 0, "weight tile instructions divide among 4 waves");

  const int nch = (a.Ktot + KC - 1) / KC;
  const int nchan = IN_HI ? a.Cv / S : a.Cv;   // real channels of the input
  TileIter it;
  it.init(a, nch);
  TileWork tw;
  while (it.next(a, nch, tw)) {
  const int c0 = tw.c0, c1 = tw.c1;
  const int rowtile = a.rt0 + tw.tile / a.ncoltiles;
  const int coltile = tw.tile % a.ncoltiles;
  const int m0 = rowtile * MB;
  const int n0 = rowtile * NPT;
  if (tile_is_dead<MB, OUT_HI>(a, m0)) continue;
  const ColTile ct = make_coltile(coltile * NB, a.Tcols, NB);

  // ---- activation descriptors, rebased to the tile's first sample (offsets stay small) ----
  const long left = (long)(a.B - ct.b0) * a.Lin * 4;
  const long nb0 = left * a.in.C0, nb1 = left * a.in.C1;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in.p0) + (size_t)ct.b0 * a.in.C0 * a.Lin, 0,
      (int)(nb0 < 0x7fffffffL ? nb0 : 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
      a.in.C1 ? const_cast<float*>(a.in.p1) + (size_t)ct.b0 * a.in.C1 * a.Lin
              : const_cast<float*>(a.in.p0),
      0, (int)(nb1 < 0x7fffffffL ? nb1 : 0x7fffffffL), 0x00020000);

  // ---- per-lane load offsets of this wave's channel: positions lane + 64*(wslice + SL*i) ----
  // (the launcher guarantees C0 == C1 whenever a tile spans several samples of two segments, so
  // one set of offsets serves both)
  int vo[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int j = lane + 64 * (wslice + SL * i);
    vo[i] = (int)0x80000000u;      // out of range: the load returns 0
    if (i < nld && j < a.RLv) {
      int s, tau;
      lds_pos_decode(ct, j, a.Tcols, a.H, s, tau);
      if (ct.b0 + s < a.B) {
        const int wq = tau + a.win_start;
        int idx = -1;
        if (IN_HI) idx = segan_hi_index(S * wq + wc, a.Lin, a.padL, a.mode, a.roll);
        else if (wq >= 0 && wq < a.Lin) idx = wq;
        if (idx >= 0) vo[i] = (s * a.in.C0 * a.Lin + idx) * 4;
      }
    }
  }
  const int il_w = wc * RLs + lane + 64 * wslice;   // LDS element of load 0

  // ---- weight tile: per-lane byte offset inside the packed buffer ----
  const int wrr = lane / F4R, wc4 = lane % F4R;
  const int wgcol = OUT_HI ? ((4 * wc4) / NPT) * a.NP + n0 + (4 * wc4) % NPT : m0 + 4 * wc4;
  const int wvo = (wrr * a.RP + wgcol) * 4;

  // ---- per-lane operand offsets ----
  int aoff[NI], boff[NJ], rsh[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rloc = 32 * (wm * NI + i);
    aoff[i] = h * MB + rloc + l31;
    rsh[i] = SHIFT ? a.rowshift[rloc / NPT] : 0;
  }
  int col_b[NJ], col_t[NJ];
  lane_columns<NB, WN, NJ>(a, ct, wn, l31, h, col_b, col_t, boff);
  // the 31-of-32-taps merges (corr_mma_chunk): operand offsets of the T form's merged steps,
  // the carried row-30 activations of the F form's channel pairs (zero: no partner seen yet)
  int aoffx[NI], boffx[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) aoffx[i] = aoff[i] + h * (U - 1) * MB;
#pragma unroll
  for (int j = 0; j < NJ; ++j) boffx[j] = boff[j] + h * (RLs - 1);
  float b30[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) b30[j] = 0.0f;

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // BLK: blocked accumulation (segan_common.h) — a second register set, so 2 instead of 3
  // waves per SIMD; without it `accs` is never touched and disappears
  f32x16 accs[BLK ? NI : 1][BLK ? NJ : 1];
  if (BLK) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[BLK ? i : 0][BLK ? j : 0][e] = 0.0f;
  }

  float ireg[NLD];
  float xsc = 1.0f, xsh = 0.0f, xsl = 1.0f;

  auto load_chunk = [&](int ch, int buf) {
    // weights: global -> LDS
    float* Wl = Wl0 + buf * (KC * MB);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int q = wave + 4 * p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrsrc, (__attribute__((address_space(3))) void*)(Wl + q * 256), 16, wvo,
          (ch * KC + q * RPI) * a.RP * 4, 0, 0);
    }
    // activations: this wave's channel of the chunk
    const int chan = IN_HI ? ch : ch * CV + wc;
    const bool cok = chan < nchan;
    const bool seg1 = chan >= a.in.C0;
    const int cc = seg1 ? chan - a.in.C0 : chan;
    const int soff = cok ? cc * a.Lin * 4 : (int)0x7fffffff;   // beyond every range: zeros
    if (XF) {
      // wave-uniform: scalar loads (the vectors were written by earlier kernels; the launcher
      // replaced NULL vectors by ones / zeros, so identity parts cost nothing special)
      typedef const __attribute__((address_space(4))) float* cptr;
      const int cx = cok ? chan : 0;
      xsl = ((cptr)a.in.slope)[cx];
      xsc = ((cptr)a.in.scale)[cx];
      xsh = ((cptr)a.in.shift)[cx];
    }
    const __amdgpu_buffer_rsrc_t rs = seg1 ? r1 : r0;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (i < nld) ireg[i] = buf_load_f32(rs, vo[i], soff);
  };
  auto store_chunk = [&](int buf) {
    float* Il = Il0 + buf * (CV * RLs) + il_w;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (i < nld) {
        float v = ireg[i];
        if (XF) {
          v = fmaf(v, xsc, xsh);
          v = v > 0.0f ? v : v * xsl;
        }
        Il[64 * SL * i] = v;
      }
    }
  };

  load_chunk(c0, 0);
  store_chunk(0);
  __syncthreads();
  // one chunk: the next chunk's loads go out, this one's MFMAs run, the next one's LDS stores follow.
  // FM / ZP select the 31-of-32-taps merges of corr_mma_chunk at COMPILE time: the variants run in
  // separate loops (or back to back inside one loop body), never behind a branch inside a loop — a
  // branch between two unrolled MFMA bodies costs a second copy of the 64 accumulator registers
  auto one_chunk = [&](auto fm, auto zp, int ch, auto sbuf) {
    // sbuf: the LDS buffer when the caller knows it statically (the pair loop), else -1
    int buf = decltype(sbuf)::value >= 0 ? decltype(sbuf)::value : (ch - c0) & 1;
    // opaque to the optimiser: with two chunks in one loop body it would otherwise keep the LDS
    // addresses of BOTH buffers in registers across the loop (+18 VGPRs: one wave per SIMD less)
    // instead of re-deriving them per chunk with a few VALU adds like the one-chunk loop does
    asm volatile("" : "+s"(buf));
    if (ch + 1 < c1) load_chunk(ch + 1, buf ^ 1);
    corr_mma_chunk<MB, U, KC, NI, NJ, SHIFT, decltype(fm)::value, decltype(zp)::value, NPT>(
        Wl0 + buf * (KC * MB), Il0 + buf * (CV * RLs), RLs, aoff, boff, rsh, acc, b30, h, aoffx, boffx);
    if (ch + 1 < c1) store_chunk(buf ^ 1);
    __syncthreads();
  };
  using Plain = IntC<0>;
  using NoZ = IntC<-1>;
  if constexpr (BLK) {
    // blocks of SEGAN_ACC_BLOCK chunks; the inner loop is the plain MFMA pipeline, the block sum is
    // folded into accs between blocks (nested loops keep the accumulators in place)
    static_assert(SEGAN_ACC_BLOCK % 2 == 0, "blocks hold whole channel pairs");
    for (int cb = c0; cb < c1; cb += SEGAN_ACC_BLOCK) {
      const int ce = min(cb + SEGAN_ACC_BLOCK, c1);
      if (IN_HI && a.f_pair) {
        // paired packing: the pairs must be contracted as pairs here too (row 31 of an odd channel
        // is not a zero row); c0, c1 and the block size are even
        for (int ch = cb; ch < ce; ch += 2) {
          one_chunk(IntC<(IN_HI && !SHIFT) ? 1 : 0>{}, NoZ{}, ch, NoZ{});
          one_chunk(IntC<(IN_HI && !SHIFT) ? 2 : 0>{}, NoZ{}, ch + 1, NoZ{});
        }
      } else {
        for (int ch = cb; ch < ce; ++ch) one_chunk(Plain{}, NoZ{}, ch, NoZ{});
      }
      acc_block_flush<NI, NJ>(acc, accs);
    }
  } else if constexpr (IN_HI && !SHIFT) {
    // F form.  K = 31 with paired channels (f_pair): every piece is whole pairs (even A, odd B) —
    // the channel count is even and the launcher keeps the stream-K cuts on even chunks
    if (a.f_pair) {
      for (int ch = c0; ch < c1; ch += 2) {
        one_chunk(IntC<1>{}, NoZ{}, ch, IntC<0>{});        // c0 is even: A in buffer 0, B in 1
        one_chunk(IntC<2>{}, NoZ{}, ch + 1, IntC<1>{});
      }
    } else {
      for (int ch = c0; ch < c1; ++ch) one_chunk(Plain{}, NoZ{}, ch, NoZ{});
    }
  } else if constexpr (OUT_HI && U == 8) {
    // T form, stride 4.  K = 31: the output phase whose tap u' = 0 is the padding tap is 2 for the
    // deconv forward (pad 13: the SHIFT variant) and 3 for the conv data gradient; any other
    // geometry runs the plain steps.  (Stride 2 — phase 1 for both — is left alone: the merged
    // steps cost the U = 16 variants 20 registers and with them the third wave per SIMD.)
    constexpr int ZPK = SHIFT ? 2 : 3;
    if (a.zphase == ZPK) {
      for (int ch = c0; ch < c1; ++ch) one_chunk(Plain{}, IntC<ZPK>{}, ch, NoZ{});
    } else {
      for (int ch = c0; ch < c1; ++ch) one_chunk(Plain{}, NoZ{}, ch, NoZ{});
    }
  } else {
    for (int ch = c0; ch < c1; ++ch) one_chunk(Plain{}, NoZ{}, ch, NoZ{});
  }
  if constexpr (BLK) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = accs[i][j];
  }

  if (tw.partial)
    slab_store<NI, NJ>(a.sk_ws + (size_t)(blockIdx.x * 2 + tw.piece) * (MB * NB), acc, tid);
  else
    corr_store_tile<MB, NB, WM, U, OUT_HI>(a, acc, m0, n0, wm, h, col_b, col_t);
  }  // tile loop
}

// Stream-K second pass: one workgroup per cut tile adds the slabs of its pieces in chunk
// order and stores the tile.  Tiles that one workgroup happened to hold whole were stored by
// the main kernel.
template <int MB, int NB, int WM, int U, bool OUT_HI, int KC>
__global__ __launch_bounds__(256) void corr_fixup_kernel(const CorrArgs a) {
  using G = TileGeom<MB, NB, WM, U>;
  constexpr int WN = G::WN, NI = G::NI, NJ = G::NJ, NPT = G::NPT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar bias loads (epi_bias)
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int nch = (a.Ktot + KC - 1) / KC;
  const int t = blockIdx.x;
  const long u0 = (long)t * nch, u1 = u0 + nch - 1;
  const int g_first = (int)(u0 / a.sk_units), g_last = (int)(u1 / a.sk_units);
  if (g_first == g_last) return;
  const int tile = a.sk_nfull + t;
  const int rowtile = a.rt0 + tile / a.ncoltiles;
  const int coltile = tile % a.ncoltiles;
  const int m0 = rowtile * MB, n0 = rowtile * NPT;
  if (tile_is_dead<MB, OUT_HI>(a, m0)) return;
  const ColTile ct = make_coltile(coltile * NB, a.Tcols, NB);
  int col_b[NJ], col_t[NJ], boff[NJ];
  lane_columns<NB, WN, NJ>(a, ct, wn, l31, h, col_b, col_t, boff);
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  // two pieces in flight at a time (the kernel is a pure stream: more loads outstanding per
  // lane), always ADDED in chunk order
  auto slab_of = [&](int g) {
    const int piece = t - (int)(((long)g * a.sk_units) / nch);
    return reinterpret_cast<const f32x4*>(a.sk_ws + (size_t)(g * 2 + piece) * (MB * NB));
  };
  constexpr int NV = NI * NJ * 4;
  for (int g = g_first; g <= g_last; g += 2) {
    const bool two = g + 1 <= g_last;
    const f32x4* s0 = slab_of(g);
    const f32x4* s1 = slab_of(two ? g + 1 : g);
    f32x4 v0[NV], v1[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) v0[q] = s0[q * 256 + tid];
    if (two) {
#pragma unroll
      for (int q = 0; q < NV; ++q) v1[q] = s1[q * 256 + tid];
    }
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q / (NJ * 4)][(q / 4) % NJ][4 * (q % 4) + e] += v0[q][e];
    if (two) {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q / (NJ * 4)][(q / 4) % NJ][4 * (q % 4) + e] += v1[q][e];
    }
  }
  corr_store_tile<MB, NB, WM, U, OUT_HI>(a, acc, m0, n0, wm, h, col_b, col_t);
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

// ====================================================================================
// conv data gradient of SHORT rows (Ls = L/S of 64 or fewer positions: the deep encoder /
// discriminator layers).  The T form above computes S*q + r over the padded row and spends
// (Ls + U-1)/Ls of its columns — 23 of 16 at S = 4, Ls = 16; 23 of 8 in the deepest stride-2
// layer — on positions that are mostly zeros.  Here the contraction is the GEMM + col2im of a
// transposed conv instead: no halo columns at all.
//   Y[(n, r, u), (b, t)] = sum_m W[m, n, S*u + r] * da[b, m, t]             (MFMA GEMM over m)
//   dxp[b, n, P]         = sum_u Y[(n, P%S, u), (b, P/S - u)]              (col2im, U terms)
//   dx[b, n, i]          = dxp[i + padL] + reflected halo terms, rolled back (as fold_halo_kernel)
// A 128 x 128 tile holds 4 channels x (S phases x U taps = 32) rows and 128/Ls samples x Ls
// positions; each wave's 64 x 64 quarter holds, for 2 channels, ALL rows of its 64/Ls samples,
// so the col2im and the reflect fold are wave-local: the quarter goes through LDS once (one
// channel at a time) and complete dx rows are stored — no halo buffer, no fold pass, no atomics.
// Both operands are K-major ([m][128 floats]) and go HBM/L2 -> LDS by LDS-DMA: the weight from
// the "G" packing Wg[m][n*32 + r*U + u] (segan_pack_weights_g), da as it lies in memory.
// ====================================================================================
#define CS_KC 16
struct ShortArgs {
  const float* da;
  const float* wg;
  float* dx;
  int B, N, M, L, Ls, padL, padRw, roll, ncoltiles;
  int xr8, xbc;   // 2-D blocked tile order inside an XCD's range: rows per XCD, block width (0: linear)
};

template <int S>
__global__ __launch_bounds__(256, 2) void conv_dgrad_short_kernel(const ShortArgs a) {
  constexpr int MB = 128, NB = 128, KC = CS_KC, NI = 2, NJ = 2, YP = 65, U = 32 / S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl0 = smem;                  // [2][KC*MB]
  float* Il0 = smem + 2 * KC * MB;   // [2][KC*NB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  // Every XCD its own range of row tiles.  Inside the range the ~128 workgroups that are resident on
  // the XCD at one time should touch as few different operand tiles as possible — each reads 512 KB
  // of Wg (its row tile) and 512 KB of da (its column tile), all of them walk the contraction in
  // step, and the XCD's 4 MB L2 holds the slices in flight: in row-major order 128 consecutive tiles
  // are 3.4 rows x 38 columns (enc4: 41 operand tiles), in blocks of (rows of the XCD) x xbc
  // columns they are 16 x 8 (24 operand tiles).  Round 5 A/B: see DESIGN.md 5.2.
  int rowtile, coltile;
  if (a.xbc > 0) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int per = a.xr8 * a.xbc;                       // tiles of a full column block
    const int cb = k / per, rem = k - cb * per;
    const int wdt = min(a.xbc, a.ncoltiles - cb * a.xbc);  // the last block may be narrower
    const int rr = rem / wdt;
    rowtile = x * a.xr8 + rr;
    coltile = cb * a.xbc + rem - rr * wdt;
  } else {
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    rowtile = vb / a.ncoltiles;
    coltile = vb - rowtile * a.ncoltiles;
  }
  const int Ls = a.Ls, L = a.L;
  const int n0 = rowtile * 4;
  const int b0 = coltile * (NB / Ls);
  const int rowpitch = a.N * 32;

  const long wbytes = (long)a.M * rowpitch * 4, xbytes = (long)a.B * a.M * Ls * 4;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wg), 0, (int)(wbytes < 0x7fffffffL ? wbytes : 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.da), 0, (int)(xbytes < 0x7fffffffL ? xbytes : 0x7fffffffL), 0x00020000);
  // one DMA instruction = 64 lanes x 16 B = two K-rows of 128 floats
  const int lr = lane >> 5, lc = 4 * (lane & 31);
  const int wvo = (lr * rowpitch + rowtile * MB + lc) * 4;
  const int s_l = lc / Ls, t_l = lc - s_l * Ls;
  const int xvo = (b0 + s_l < a.B) ? ((s_l * a.M + lr) * Ls + t_l) * 4 : (int)0x80000000u;

  auto load_chunk = [&](int ch, int buf) {
    float* Wl = Wl0 + buf * (KC * MB);
    float* Il = Il0 + buf * (KC * NB);
#pragma unroll
    for (int p = 0; p < KC / 8; ++p) {
      const int q = wave + 4 * p;       // K-rows 2q, 2q+1 of the chunk
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrs, (__attribute__((address_space(3))) void*)(Wl + q * 256), 16, wvo,
          (ch * KC + 2 * q) * rowpitch * 4, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          xrs, (__attribute__((address_space(3))) void*)(Il + q * 256), 16, xvo,
          ((b0 * a.M + ch * KC + 2 * q) * Ls) * 4, 0, 0);
    }
  };

  int aoff[NI], boff[NJ], rsh[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    aoff[i] = h * MB + 32 * (wm * NI + i) + l31;
    rsh[i] = 0;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) boff[j] = h * NB + 64 * wn + 32 * j + l31;

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nch = a.M / KC;
  float b30_unused[NJ] = {};
  load_chunk(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the DMA of chunk 0 has landed
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) load_chunk(ch + 1, buf ^ 1);
    corr_mma_chunk<MB, 1, KC, NI, NJ, false>(Wl0 + buf * (KC * MB), Il0 + buf * (KC * NB), NB, aoff,
                                             boff, rsh, acc, b30_unused, h, aoff, boff);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
  }

  // ---- col2im + reflect fold, one channel of the wave's quarter at a time ----
  float* Yb = smem + wave * (32 * YP);     // [32 rows (r, u)][64 columns], row pitch 65
  const int bw = b0 + (64 * wn) / Ls;      // first sample of this wave's columns
  // dxp[P] of local sample sl from the staged rows
  auto dxp = [&](int P, int cbase) -> float {
    const int r = P % S, qq = P / S;
    float v = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = qq - u;
      if (t >= 0 && t < Ls) v += Yb[(r * U + u) * YP + cbase + t];
    }
    return v;
  };
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Yb[((e & 3) + 8 * (e >> 2) + 4 * h) * YP + 32 * j + l31] = acc[i][j][e];
    __syncthreads();
    const int n = n0 + 2 * wm + i;
#pragma unroll
    for (int p = 0; p < S; ++p) {
      const int o = p * 64 + lane;         // 64*S outputs: 64/Ls samples x L positions
      const int sl = o / L, io = o - sl * L;
      const int b = bw + sl;
      float v = dxp(io + a.padL, sl * Ls);
      if (io >= 1 && io <= a.padL) v += dxp(a.padL - io, sl * Ls);
      const int jr = L - 2 - io;           // right halo sample that mirrors onto io
      if (jr >= 0 && jr < a.padRw) v += dxp(L + a.padL + jr, sl * Ls);
      if (b < a.B && n < a.N) {
        int ii = io - a.roll;
        if (ii < 0) ii += L;
        if (ii >= L) ii -= L;
        a.dx[((size_t)b * a.N + n) * L + ii] = v;
      }
    }
    __syncthreads();
  }
}

__global__ void pack_g_kernel(const float* __restrict__ w, float* __restrict__ wg, int M, int N,
                              int K, int S) {
  const size_t total = (size_t)M * N * 32;
  const int U = 32 / S;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i & 31);           // r*U + u
    const size_t mn = i >> 5;
    const int k = S * (c % U) + c / U;
    wg[i] = k < K ? w[mn * K + k] : 0.0f;
  }
}

// diagnostics: how the last forward / data-gradient contraction of this thread was launched
static thread_local int g_last_corr[6];
void segan_note_corr_launch(int kind, unsigned grid, const CorrArgs& a, int ntiles) {
  g_last_corr[0] = kind;             // 1 corr_kernel, 2 corr2_kernel, 3 corr_bf2_kernel (bf16 / bf16x3)
  g_last_corr[1] = (int)grid;
  g_last_corr[2] = ntiles;
  g_last_corr[3] = a.sk_nfull;       // tiles run whole; the other ntiles - sk_nfull were cut (stream-K)
  g_last_corr[4] = a.sk_total > 0 ? a.sk_units : 0;
  g_last_corr[5] = a.xf_mode;
}
extern "C" void segan_debug_last_corr(int* out6) {
  for (int i = 0; i < 6; ++i) out6[i] = g_last_corr[i];
}

// ---- launch plumbing ---------------------------------------------------------------------
// One process drives one GPU, but nothing here assumes it: what is cached is cached per device.
static int cur_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d & 15;
}

template <typename K>
static void allow_big_lds(K kern, bool (&done)[16]) {
  const int d = cur_device();
  if (!done[d]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done[d] = true;
  }
}

// workgroups resident per CU (registers / LDS), cached per device and LDS size
template <typename K>
static int resident_per_cu(K kern, size_t lds, int (&occ)[16], size_t (&occ_lds)[16]) {
  const int d = cur_device();
  if (occ[d] == 0 || occ_lds[d] != lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256,
                                                     lds) != hipSuccess || nb < 1)
      nb = 2;
    occ[d] = nb > 4 ? 4 : nb;
    occ_lds[d] = lds;
  }
  return occ[d];
}

// Data-parallel + stream-K plan of a launch of `ntiles` equal tiles of `nch` chunks.  All
// workgroups do the same amount of work and several are resident per CU, so a one-tile-per-
// workgroup launch takes ceil(tiles / resident) rounds; when the last round would leave more
// than 3 % of the CU-time idle, whole rounds are run tile-per-workgroup and the tiles of the
// last partial round are cut along the contraction across all workgroups (slabs in the
// caller's scratch + corr_fixup_kernel; without enough scratch the launch stays classic).
static unsigned plan_streamk(CorrArgs& a, int ntiles, int nch, int occ, size_t slab_floats,
                             bool allow, bool pair_cuts = false) {
  a.sk_nfull = ntiles;
  a.sk_units = 1;
  a.sk_total = 0;
  const double classic_eff = (double)ntiles / (256.0 * ceil_div(ntiles, 256));
  if (!allow || a.act != SEGAN_ACT_NONE || ntiles < 64 || nch < 8 || classic_eff >= 0.97) {
    // no cut tiles.  Two or more rounds still run on PERSISTENT workgroups (one per resident slot,
    // striding over the whole tiles) rather than one workgroup per tile: measured (round 4,
    // alternating runs) +3-4 % on launches of short tiles — enc1's data gradient, 4834 tiles of 64
    // chunks: 1.35 -> 1.31 ms — and nothing on long ones (dec3: 4800 tiles of 128 chunks)
    if (allow && a.act == SEGAN_ACT_NONE && ntiles >= 2 * 256 * occ) return (unsigned)segan_grid_slots(occ);
    return (unsigned)ntiles;
  }
  const int G = segan_grid_slots(occ);
  if (a.sk_ws == nullptr || a.sk_ws_floats < (size_t)G * 2 * slab_floats) return (unsigned)ntiles;
  a.sk_nfull = (ntiles / G) * G;
  const int rem = ntiles - a.sk_nfull;
  if (rem == 0) return (unsigned)ntiles;
  a.sk_total = (long)rem * nch;
  a.sk_units = (int)((a.sk_total + G - 1) / G);
  // paired channels (corr2_kernel with f_pair: a chunk is ONE channel, nch = the even channel
  // count): cuts on even chunks only, so that every piece is whole (A, B) pairs.  corr_kernel does
  // not pair (it zeroes the borrowed row while staging and runs the plain 16 steps), its pieces may
  // be odd (round-5 advice)
  if (pair_cuts) a.sk_units += a.sk_units & 1;
  return (unsigned)G;
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
  static bool attr_done[16];
  static int occ[16];
  static size_t occ_lds[16];
  allow_big_lds(kern, attr_done);
  // rows below rt0 all go to a NULL destination (the z half of the first decoder layer)
  a.rt0 = (!OUT_HI && a.out0 == nullptr) ? a.OC0 / MB : 0;
  const int ntiles = (nrowtiles - a.rt0) * a.ncoltiles;
  const int nch = ceil_div(a.Ktot, KC);
  const unsigned grid = plan_streamk(a, ntiles, nch, resident_per_cu(kern, lds, occ, occ_lds),
                                     (size_t)MB * NB, allow_sk);
  segan_note_corr_launch(1, grid, a, ntiles);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  if (int e = segan_check_launch("corr_kernel")) return e;
  if (a.sk_total > 0) {
    hipLaunchKernelGGL((corr_fixup_kernel<MB, NB, WM, U, OUT_HI, KC>), dim3(ntiles - a.sk_nfull),
                       dim3(256), 0, st, a);
    return segan_check_launch("corr_fixup_kernel");
  }
  return SEGAN_OK;
}

template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, bool SHIFT, bool XF, bool BLK = false>
static int launch_corr2_x(CorrArgs a, hipStream_t st, bool allow_sk) {
  constexpr int KC = CORR2_KC;
  constexpr int CV = KC / U;
  constexpr int S = 32 / U;
  constexpr int SL = 4 / CV;
  // T form: NPT = MB / S channels per tile out of the NP the packing pads to (a multiple of 128 / S:
  // the small-row tiles below it leave the all-padding row tiles out)
  const int nrowtiles = OUT_HI ? ceil_div(a.Nout, MB / S) : ceil_div(a.Rvalid, MB);
  a.RLv = a.RLs;
  a.RLs = round_up(a.RLs, 64 * SL);
  a.nld = a.RLs / (64 * SL);
  const size_t lds = (size_t)(2 * KC * MB + 2 * CV * a.RLs) * sizeof(float);
  auto kern = corr2_kernel<MB, NB, WM, U, IN_HI, OUT_HI, SHIFT, XF, BLK>;
  static bool attr_done[16];
  static int occ[16];
  static size_t occ_lds[16];
  allow_big_lds(kern, attr_done);
  a.rt0 = (!OUT_HI && a.out0 == nullptr) ? a.OC0 / MB : 0;
  const int ntiles = (nrowtiles - a.rt0) * a.ncoltiles;
  const int nch = ceil_div(a.Ktot, KC);
  const unsigned grid = plan_streamk(a, ntiles, nch, resident_per_cu(kern, lds, occ, occ_lds),
                                     (size_t)MB * NB, allow_sk, a.f_pair != 0);
  // the pair loop of corr2_kernel runs one_chunk(ch) and one_chunk(ch + 1) unconditionally: every
  // piece it is handed must start and end on an even chunk
  SEGAN_REQUIRE(!a.f_pair || ((nch & 1) == 0 && (a.sk_total == 0 || (a.sk_units & 1) == 0)),
                "corr2: paired channels need even pieces (nch=%d, sk_units=%d)", nch, a.sk_units);
  segan_note_corr_launch(2, grid, a, ntiles);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  if (int e = segan_check_launch("corr2_kernel")) return e;
  if (a.sk_total > 0) {
    hipLaunchKernelGGL((corr_fixup_kernel<MB, NB, WM, U, OUT_HI, KC>), dim3(ntiles - a.sk_nfull),
                       dim3(256), 0, st, a);
    return segan_check_launch("corr_fixup_kernel");
  }
  return SEGAN_OK;
}

template <int MB, int NB, int WM, int U, bool IN_HI, bool OUT_HI, bool SHIFT>
static int launch_corr2_t(CorrArgs& a, hipStream_t st, bool allow_sk) {
  // blocked accumulation: built for the stride-4 (U = 8) 128-row tiles, the long contractions
  // of the SEGAN+ nets; elsewhere the plain kernels run (contractions a quarter as long)
  if constexpr (U == 8 && MB == 128) {
    if (a.acc_block)
      return a.xf_mode ? launch_corr2_x<MB, NB, WM, U, IN_HI, OUT_HI, SHIFT, true, true>(a, st, allow_sk)
                       : launch_corr2_x<MB, NB, WM, U, IN_HI, OUT_HI, SHIFT, false, true>(a, st, allow_sk);
  }
  return a.xf_mode ? launch_corr2_x<MB, NB, WM, U, IN_HI, OUT_HI, SHIFT, true>(a, st, allow_sk)
                   : launch_corr2_x<MB, NB, WM, U, IN_HI, OUT_HI, SHIFT, false>(a, st, allow_sk);
}

// corr2_kernel covers a launch when its window fits 4 loads per lane, a padded zero stays a
// zero under the input transform, and both channel segments have the same sample pitch
// wherever a tile spans several samples (corr_kernel handles the rest)
template <int U>
static bool corr2_ok(const CorrArgs& a, int NB) {
  constexpr int SL = 4 / (CORR2_KC / U);
  if (a.RLs > 64 * SL * CORR2_NLD) return false;
  if (a.mode == SEGAN_PAD_ZERO && a.xf_mode == 2) return false;
  if (a.in.C1 > 0 && a.in.C0 != a.in.C1 && samples_per_tile(a.Tcols, NB) > 1) return false;
  return true;
}

// ---- F form (conv forward, deconv data gradient) ----
template <int U>
static int launch_corr_f(CorrArgs& a, hipStream_t st) {
  constexpr int NB = 128;
  a.ncoltiles = ceil_div(a.Ctot, NB);
  a.RLs = NB + samples_per_tile(a.Tcols, NB) * a.H;
  const bool small = a.Rvalid <= 64;
  if constexpr (U == 16) {
    // stride 2 with at most 32 output rows (the 16 / 32-channel layers of the 11-layer stride-2 shape:
    // round-5 review, weak 3): 32 x 256 tiles, waves 1 x 4 — a 64-row tile would be half padding
    if (a.Rvalid <= 32) {
      constexpr int NBW = 256;
      CorrArgs b = a;
      b.ncoltiles = ceil_div(b.Ctot, NBW);
      b.RLs = NBW + samples_per_tile(b.Tcols, NBW) * b.H;
      if (corr2_ok<U>(b, NBW)) {
        a = b;
        return launch_corr2_t<32, NBW, 1, U, true, false, false>(a, st, false);
      }
    }
  }
  if (corr2_ok<U>(a, NB))
    return small ? launch_corr2_t<64, NB, 2, U, true, false, false>(a, st, false)
                 : launch_corr2_t<128, NB, 2, U, true, false, false>(a, st, true);
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
  constexpr int NB = 128;
  a.ncoltiles = ceil_div(a.Ctot, NB);
  a.RLs = NB + samples_per_tile(a.Tcols, NB) * a.H;
  if (a.RLs > 512) {
    segan_set_error("corr: sample length %d too short for stride %d (RLs=%d)", a.Tcols, 32 / U,
                    a.RLs);
    return SEGAN_EUNSUPPORTED;
  }
  if (corr2_ok<U>(a, NB)) {
    if constexpr (U == 16) {
      // stride 2, few output channels (the shallow layers of the 11-layer stride-2 shape): a 128-row
      // tile is 64 channels x 2 phases — with 32 / 16 channels half / three quarters of its MFMAs
      // multiply padding.  64-row tiles (32 channels), and 32-row tiles (16 channels, both phases
      // inside one 32-row MFMA block: only without row shifts, i.e. for the conv data gradient)
      // ... on 256 columns for the transposed-conv forward and for the 16-channel tile: twice the
      // columns halve what the per-tile prologue / epilogue weigh.  Measured per layer on one box,
      // alternating (round 6): deconv forward of 32 / 32 / 16 channels 0.611 / 0.608 / 0.635 -> 0.593 /
      // 0.598 / 0.598 ms, conv data gradient into 16 channels 0.39 -> 0.37 ms, into 32 channels 0.36 ->
      // 0.37 - 0.38 ms (worse: those keep 128 columns).  SEGAN_T_WIDE=0: 128 columns everywhere
      static const bool wide = [] { const char* e = getenv("SEGAN_T_WIDE"); return !(e && e[0] == '0'); }();
      if (a.Nout <= 32 && wide && (a.halo == nullptr || a.Nout <= 16)) {
        constexpr int NBW = 256;
        CorrArgs b = a;
        b.ncoltiles = ceil_div(b.Ctot, NBW);
        b.RLs = NBW + samples_per_tile(b.Tcols, NBW) * b.H;
        if (corr2_ok<U>(b, NBW)) {
          a = b;
          if (!SHIFT && a.Nout <= 16) return launch_corr2_t<32, NBW, 1, U, false, true, false>(a, st, true);
          return launch_corr2_t<64, NBW, 1, U, false, true, SHIFT>(a, st, true);
        }
      }
      if (!SHIFT && a.Nout <= 16) return launch_corr2_t<32, NB, 1, U, false, true, false>(a, st, true);
      if (a.Nout <= 32) return launch_corr2_t<64, NB, 1, U, false, true, SHIFT>(a, st, true);
    }
    return launch_corr2_t<128, NB, 1, U, false, true, SHIFT>(a, st, true);
  }
  constexpr int KC = U <= 16 ? 32 : KCH;
  if (a.RLs <= 256) return launch_corr_t<128, NB, 1, U, false, true, SHIFT, 1, KC>(a, st, true);
  return launch_corr_t<128, NB, 1, U, false, true, SHIFT, 2, KCH>(a, st, true);
}

template <bool IN_HI, bool OUT_HI>
static int launch_corr(CorrArgs& a, int U, hipStream_t st) {
  a.xf_mode = a.in.shift ? 2 : ((a.in.scale || a.in.slope) ? 1 : 0);
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

static void set_scratch(CorrArgs& a, void* scratch, size_t scratch_bytes) {
  a.sk_ws = (float*)scratch;
  a.sk_ws_floats = scratch ? scratch_bytes / sizeof(float) : 0;
}

// ====================================================================================
// C ABI: forward and data-gradient entry points
// ====================================================================================
// scratch of the bf16 / bf16x3 forms of the four entry points below: the packed activation
// operand.  op: 0 conv forward, 1 conv data gradient, 2 deconv forward, 3 deconv data gradient;
// (N, M) as in the entry point's weight [M, N, K] convention, L = length of the HIGH-rate side.
extern "C" size_t segan_bf16_scratch_bytes(int op, int B, int N, int M, int L, int K, int S, int pad,
                                           int planes) {
  if (!stride_ok(S) || B <= 0 || N <= 0 || M <= 0 || L <= 0 || L % S != 0 || planes < 1 || planes > 3)
    return 0;
  const int U = 32 / S, Ls = L / S;
  // + the stream-K slabs (as segan_corr_scratch_bytes) behind the packed operand
  const size_t slabs = 256 + (size_t)1024 * 2 * 128 * 128 * sizeof(float);
  switch (op) {
    case 0: return slabs + segan_corr_bf2_scratch_bytes(B, N * S, Ls, U - 1, planes);
    case 3: return slabs + segan_corr_bf2_scratch_bytes(B, N * S, Ls, U - 1, planes);
    case 2: {
      int cmin = 1 << 30, cmax = 0;
      for (int r = 0; r < S; ++r) {
        const int c = (r + pad) / S;
        cmin = c < cmin ? c : cmin;
        cmax = c > cmax ? c : cmax;
      }
      return slabs + segan_corr_bf2_scratch_bytes(B, M, Ls, U - 1 + (cmax - cmin), planes);
    }
    case 1: {
      const int padR = K - S - pad > 0 ? K - S - pad : 0;
      return slabs + segan_corr_bf2_scratch_bytes(B, M, (L + pad + padR - 1) / S + 1, U - 1, planes);
    }
  }
  return 0;
}

extern "C" size_t segan_corr_scratch_bytes(void) {
  // 256 CUs x 4 resident workgroups x 2 slabs of a 128 x 128 fp32 tile
  return (size_t)1024 * 2 * 128 * 128 * sizeof(float);
}

extern "C" int segan_conv1d_fwd(const segan_src* x, const void* wf, const float* bias, float* out,
                                int B, int N, int M, int L, int K, int S, int padL, int mode,
                                int roll, int precision, void* scratch, size_t scratch_bytes,
                                void* stream) {
  const int acc_block = precision == SEGAN_PREC_FP32_BLOCKED;
  if (acc_block) precision = SEGAN_PREC_FP32;
  SEGAN_REQUIRE(precision_ok(precision), "conv1d_fwd: precision must be 0, 1, 3 or 4");
  SEGAN_REQUIRE(stride_ok(S), "conv1d_fwd: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "conv1d_fwd: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && L > 0, "conv1d_fwd: bad sizes");
  SEGAN_REQUIRE(L % S == 0, "conv1d_fwd: length %d not divisible by stride %d", L, S);
  SEGAN_REQUIRE(wf && out, "conv1d_fwd: NULL pointer");
  SEGAN_REQUIRE(mode == SEGAN_PAD_REFLECT || mode == SEGAN_PAD_ZERO, "conv1d_fwd: bad pad mode");
  // right over-run of the last output's window: L - S + K - 1 - padL - (L - 1) = K - S - padL
  SEGAN_REQUIRE(mode != SEGAN_PAD_REFLECT || (padL < L && K - S - padL < L),
                "conv1d_fwd: reflect padding %d needs length > pad (L=%d)", padL, L);
  SEGAN_REQUIRE(roll > -L && roll < L, "conv1d_fwd: |roll| must be < L");
  if (int e = check_src(x, N, "conv1d_fwd")) return e;
  const int U = 32 / S;
  CorrArgs a = {};
  a.acc_block = acc_block;
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
  a.f_pair = f_pair(N, K); a.zphase = -1;
  if (precision) {
    CorrArgs a2 = a;
    const int e = segan_corr_bf2_f(a2, U, wf, precision, scratch, scratch_bytes, (hipStream_t)stream);
    return e;      // SEGAN_EUNSUPPORTED: the caller runs the fp32 form of this geometry
  }
  if (N <= 2) return segan_launch_fsmall(a, M, N, S, (hipStream_t)stream);
  set_scratch(a, scratch, scratch_bytes);
  return launch_corr<true, false>(a, U, (hipStream_t)stream);
}

extern "C" int segan_deconv1d_dgrad(const float* dy, const void* wf, float* dx0, float* dx1, int B,
                                    int M, int M0, int N, int Ls, int K, int S, int pad,
                                    int precision, void* scratch, size_t scratch_bytes,
                                    void* stream) {
  const int acc_block = precision == SEGAN_PREC_FP32_BLOCKED;
  if (acc_block) precision = SEGAN_PREC_FP32;
  SEGAN_REQUIRE(precision_ok(precision), "deconv1d_dgrad: precision must be 0, 1, 3 or 4");
  SEGAN_REQUIRE(stride_ok(S), "deconv1d_dgrad: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "deconv1d_dgrad: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && Ls > 0, "deconv1d_dgrad: bad sizes");
  SEGAN_REQUIRE(M0 >= 0 && M0 <= M, "deconv1d_dgrad: split %d outside [0,%d]", M0, M);
  SEGAN_REQUIRE(dy && wf, "deconv1d_dgrad: NULL pointer");
  SEGAN_REQUIRE(dx0 || dx1, "deconv1d_dgrad: both destinations NULL");
  const int U = 32 / S;
  CorrArgs a = {};
  a.acc_block = acc_block;
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
  a.f_pair = f_pair(N, K); a.zphase = -1;
  if (precision) {
    CorrArgs a2 = a;
    const int e = segan_corr_bf2_f(a2, U, wf, precision, scratch, scratch_bytes, (hipStream_t)stream);
    return e;      // SEGAN_EUNSUPPORTED: the caller runs the fp32 form of this geometry
  }
  set_scratch(a, scratch, scratch_bytes);
  return launch_corr<true, false>(a, U, (hipStream_t)stream);
}

extern "C" int segan_deconv1d_fwd(const segan_src* x, const void* wt, const float* w,
                                  const float* bias, float* y, int B, int M, int N, int Ls, int K,
                                  int S, int pad, int act, int precision, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  const int acc_block = precision == SEGAN_PREC_FP32_BLOCKED;
  if (acc_block) precision = SEGAN_PREC_FP32;
  SEGAN_REQUIRE(precision_ok(precision), "deconv1d_fwd: precision must be 0, 1, 3 or 4");
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
  a.acc_block = acc_block;
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
  a.f_pair = 0; a.zphase = t_zphase(K, S, pad);
  if (w && N <= 2) return segan_launch_tsmall(a, w, K, M, N, S, pad, (hipStream_t)stream);
  if (precision && act == SEGAN_ACT_NONE) {
    CorrArgs a2 = a;
    const int e = segan_corr_bf2_t(a2, U, wt, precision, scratch, scratch_bytes, (hipStream_t)stream);
    return e;      // SEGAN_EUNSUPPORTED: the caller runs the fp32 form of this geometry
  }
  SEGAN_REQUIRE(precision == 0, "deconv1d_fwd: tanh epilogue only on the fp32 path");
  set_scratch(a, scratch, scratch_bytes);
  return launch_corr<false, true>(a, U, (hipStream_t)stream);
}

extern "C" int segan_conv1d_dgrad(const float* da, const void* wt, const float* w, float* dx,
                                  float* halo, int B, int N, int M, int L, int K, int S, int padL,
                                  int roll, int precision, void* scratch, size_t scratch_bytes,
                                  void* stream) {
  const int acc_block = precision == SEGAN_PREC_FP32_BLOCKED;
  if (acc_block) precision = SEGAN_PREC_FP32;
  SEGAN_REQUIRE(precision_ok(precision), "conv1d_dgrad: precision must be 0, 1, 3 or 4");
  SEGAN_REQUIRE(stride_ok(S), "conv1d_dgrad: stride %d not in {1,2,4}", S);
  SEGAN_REQUIRE(K >= 1 && K <= 32, "conv1d_dgrad: kernel width %d not in [1,32]", K);
  SEGAN_REQUIRE(B > 0 && N > 0 && M > 0 && L > 0 && L % S == 0, "conv1d_dgrad: bad sizes");
  SEGAN_REQUIRE(da && dx && halo && (wt || (w && N <= 2)), "conv1d_dgrad: NULL pointer");
  SEGAN_REQUIRE(roll > -L && roll < L, "conv1d_dgrad: |roll| must be < L");
  // the padded input spans K-1 halo positions, but the last S-1 of the right halo are beyond the
  // window of the last output (P <= S*(Ls-1) + K-1): they get no gradient, are neither computed
  // nor folded, and the column range of the contraction ends one phase block earlier
  const int padR = K - 1 - padL >= 0 ? (K - S - padL > 0 ? K - S - padL : 0) : -1;
  SEGAN_REQUIRE(padL >= 0 && padR >= 0 && padL < L && K - S - padL < L, "conv1d_dgrad: bad padding");
  const int U = 32 / S;
  const int Ls = L / S;
  hipStream_t st = (hipStream_t)stream;
  CorrArgs a = {};
  a.acc_block = acc_block;
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
  a.f_pair = 0; a.zphase = t_zphase(K, S, 0);      // the conv's Wt is packed for pad_t = 0
  set_scratch(a, scratch, scratch_bytes);
  int e;
  if (w && N <= 2) {
    e = segan_launch_tsmall(a, w, K, M, N, S, 0, st);
  } else if (precision) {
    CorrArgs a2 = a;
    e = segan_corr_bf2_t(a2, U, wt, precision, scratch, scratch_bytes, st);      // may decline
  } else {
    e = launch_corr<false, true>(a, U, st);
  }
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

// ---- short rows (GEMM + col2im form) -------------------------------------------------------
static bool short_ok(int N, int M, int L, int K, int S, int padL) {
  if (!stride_ok(S) || L % S != 0) return false;
  const int Ls = L / S;
  return (Ls == 4 || Ls == 8 || Ls == 16 || Ls == 32 || Ls == 64) && N % 4 == 0 &&
         M % CS_KC == 0 && K >= 1 && K <= 32 && padL >= 0 && K - 1 - padL >= 0 && padL < L &&
         K - S - padL < L;
}

extern "C" size_t segan_packed_g_bytes(int M, int N, int S) {
  if (!stride_ok(S) || M <= 0 || N <= 0) return 0;
  return (size_t)M * N * 32 * sizeof(float);
}

extern "C" int segan_pack_weights_g(const float* w, float* wg, int M, int N, int K, int S,
                                    void* stream) {
  SEGAN_REQUIRE(w && wg && M > 0 && N > 0, "pack_weights_g: bad arguments");
  SEGAN_REQUIRE(stride_ok(S) && K >= 1 && K <= 32, "pack_weights_g: stride in {1,2,4}, K <= 32");
  const size_t total = (size_t)M * N * 32;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_g_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wg, M, N, K,
                     S);
  return segan_check_launch("pack_weights_g");
}

extern "C" int segan_conv1d_dgrad_short(const float* da, const float* wg, float* dx, int B, int N,
                                        int M, int L, int K, int S, int padL, int roll,
                                        void* stream) {
  SEGAN_REQUIRE(da && wg && dx && B > 0, "conv1d_dgrad_short: bad arguments");
  SEGAN_REQUIRE(roll > -L && roll < L, "conv1d_dgrad_short: |roll| must be < L");
  if (!short_ok(N, M, L, K, S, padL)) {
    segan_set_error("conv1d_dgrad_short: geometry not covered (L/S in {4,8,16,32,64}, N %% 4 == 0, "
                    "M %% 16 == 0)");
    return SEGAN_EUNSUPPORTED;
  }
  ShortArgs a;
  a.da = da; a.wg = wg; a.dx = dx;
  a.B = B; a.N = N; a.M = M; a.L = L; a.Ls = L / S; a.padL = padL;
  a.padRw = K - S - padL > 0 ? K - S - padL : 0;
  a.roll = roll;
  a.ncoltiles = ceil_div(B, 128 / a.Ls);
  const size_t lds = 4 * 32 * 65 * sizeof(float);      // >= 2 * CS_KC * 256 floats of the main loop
  const dim3 grid((unsigned)(N / 4 * a.ncoltiles));
  // 2-D blocked order when the row tiles divide among the 8 XCDs (SEGAN_SHORT_ORDER=0: row-major)
  static const bool blocked = [] {
    const char* e = getenv("SEGAN_SHORT_ORDER");
    return !(e && e[0] == '0');
  }();
  a.xr8 = a.xbc = 0;
  if (blocked && (N / 4) % 8 == 0) {
    a.xr8 = N / 4 / 8;
    a.xbc = a.xr8 >= 128 ? 1 : 128 / a.xr8;
  }
  hipStream_t st = (hipStream_t)stream;
  if (S == 4) hipLaunchKernelGGL(conv_dgrad_short_kernel<4>, grid, dim3(256), lds, st, a);
  else if (S == 2) hipLaunchKernelGGL(conv_dgrad_short_kernel<2>, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(conv_dgrad_short_kernel<1>, grid, dim3(256), lds, st, a);
  return segan_check_launch("conv_dgrad_short_kernel");
}
