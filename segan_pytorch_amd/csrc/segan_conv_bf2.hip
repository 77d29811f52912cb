// segan_conv_bf2.hip — round-3 form of the F and T contractions on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16): BOTH operands reach LDS by LDS-DMA, nothing is converted, masked or
// transformed inside the hot loop.
//
// Why: a bf16 MFMA retires 16x the products of the fp32 one in half its cycles, so staging work
// that costs the fp32 kernels 0.5 VALU per MFMA would cost these 16 per MFMA.  Round 1's
// corr_bf_kernel converted fp32 -> bf16 while staging (8-12 VALU, 0.6 VMEM, 0.3 ds_write_b128 per
// MFMA: 0.15-0.24 of the bf16 peak).  Here the activation operand is converted ONCE per call by
// act_pack_kernel — transform (BatchNorm scale / shift, PReLU slope, alpha), both torch.cat
// segments, reflect / zero padding, the discriminator's roll, the polyphase split and the halo
// are all applied there — into the exact piece order the tile wants:
//
//   P[plane][b][g][j][8]   bf16, g = group of 8 virtual channels, j = window position
//                          F form: virtual channel (n, r), value xpad[n, S*(j + win_start) + r]
//                          T form: virtual channel m,      value x[m, j + win_start] (0 outside)
//
// so a tile's operand for one 16-channel group is, per half g, a run of consecutive 16-byte
// pieces per sample: one buffer_load_dwordx4 ... lds per 64 positions, per-lane offsets constant
// for the whole tile, the channel group in the scalar offset.  The packed weights were already
// stored in tile order (segan_pack_weights_bf) and stream the same way.  Per stage of TU taps a
// wave issues 4 (weights) + ~0.4 (activations) DMA instructions, TU * (NI + NJ) ds_read_b128 and
// TU * NI * NJ MFMAs — and nothing else.
//
// The packing pass is HBM-bound (4 B read, 2 B written per element and plane, coalesced both
// ways).  Tile geometry, stream-K hybrid and epilogues are those of corr_bf_kernel.
#include "segan_conv_shared.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3b(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

// ====================================================================================
// activation packing
// ====================================================================================
struct PackArgs {
  segan_src in;
  __bf16* out;
  size_t plane_elems;     // elements between planes
  int B, Cv, G, Qp;       // virtual channels, groups of 8 (even), positions per row
  int Lin, padL, mode, roll, win_start;
  int identity;
};

// one thread = one 16-byte piece (b, g, j); lanes run along j (coalesced reads of each of the
// piece's rows, coalesced 16-byte writes).  F form: a piece holds 8/S real channels x S phases,
// and the S padded samples S*wq .. S*wq + S-1 of a channel are consecutive in memory except at
// the reflected ends and the roll's wrap point: one vector load per channel there.
template <int S, bool IN_HI, int NPL>
__global__ __launch_bounds__(256) void act_pack_kernel(const PackArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, b = blockIdx.z;
  if (j >= a.Qp) return;
  const int wq = j + a.win_start;
  float v[8];
  if (IN_HI) {
    int idx[S];
#pragma unroll
    for (int r = 0; r < S; ++r) idx[r] = segan_hi_index(S * wq + r, a.Lin, a.padL, a.mode, a.roll);
    bool run = idx[0] >= 0;
#pragma unroll
    for (int r = 1; r < S; ++r) run = run && idx[r] == idx[0] + r;
#pragma unroll
    for (int c = 0; c < 8 / S; ++c) {
      const int n = (8 * g) / S + c;
      const bool cok = n * S < a.Cv;
      const float* row = segan_src_row(a.in, b, cok ? n : 0, a.Lin);
      float t[S];
      if (run) {
        if (S == 4) {
          typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
          const f4u x4 = *reinterpret_cast<const f4u*>(row + idx[0]);
          t[0] = x4[0]; t[1 % S] = x4[1]; t[2 % S] = x4[2]; t[3 % S] = x4[3];
        } else if (S == 2) {
          typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
          const f2u x2 = *reinterpret_cast<const f2u*>(row + idx[0]);
          t[0] = x2[0]; t[1 % S] = x2[1];
        } else {
          t[0] = row[idx[0]];
        }
      } else {
#pragma unroll
        for (int r = 0; r < S; ++r) t[r] = idx[r] >= 0 ? row[idx[r]] : 0.0f;
      }
      ChanXf xf;
      if (!a.identity) xf = segan_chan_xf(a.in, cok ? n : 0);
#pragma unroll
      for (int r = 0; r < S; ++r) {
        float x = t[r];
        if (!a.identity) x = segan_apply_xf(xf, x);
        v[c * S + r] = (cok && (run || idx[r] >= 0)) ? x : 0.0f;
      }
    }
  } else {
    const bool pok = wq >= 0 && wq < a.Lin;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cv = 8 * g + e;
      const bool ok = pok && cv < a.Cv;
      float x = 0.0f;
      if (ok) {
        x = segan_src_row(a.in, b, cv, a.Lin)[wq];
        if (!a.identity) x = segan_apply_xf(segan_chan_xf(a.in, cv), x);
      }
      v[e] = x;
    }
  }
  bf16x8 pl[3];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 p1, p2, p3;
    split3b(v[e], p1, p2, p3);
    pl[0][e] = p1; pl[1][e] = p2; pl[2][e] = p3;
  }
  const size_t piece = ((size_t)b * a.G + g) * a.Qp + j;
#pragma unroll
  for (int p = 0; p < NPL; ++p)
    *reinterpret_cast<u32x4*>(a.out + p * a.plane_elems + piece * 8) = __builtin_bit_cast(u32x4, pl[p]);
}

// ---- tile epilogue: bias, store (shared by the contraction kernel and the stream-K fix-up) ----
template <int MB, int NB, int WM, int U, bool OUT_HI>
__device__ __forceinline__ void bf2_store_tile(const CorrArgs& a,
                                               const f32x16 (&acc)[MB / (32 * WM)][NB / (32 * (4 / WM))],
                                               int m0, int n0, int wm, int h,
                                               const int (&col_b)[NB / (32 * (4 / WM))],
                                               const int (&col_t)[NB / (32 * (4 / WM))]) {
  constexpr int S = 32 / U;
  constexpr int WN = 4 / WM;
  constexpr int NI = MB / (32 * WM);
  constexpr int NJ = NB / (32 * WN);
  constexpr int NPT = MB / S;
  constexpr bool add_bias = true;
  // ---- epilogue ----
  // A bf16 tile spends 16x fewer matrix-pipe cycles per output element than an fp32 one, so the
  // generic epilogue below (64-bit address arithmetic, row / destination tests and a bias load
  // per element: ~30 VALU per stored value) cost as much as the tile's whole contraction.  Full
  // interior tiles — all rows valid and in one destination, plain stores — take the fast form:
  // buffer stores whose per-lane offsets are computed once per column (masked columns carry an
  // out-of-range offset: the store is dropped), the row inside the tile goes through the scalar
  // offset, no VALU per element.
  if (!OUT_HI) {
    float* fdst = m0 < a.OC0 ? a.out0 : a.out1;
    const int foc = m0 < a.OC0 ? a.OC0 : a.OC1;
    const int foch = m0 < a.OC0 ? m0 : m0 - a.OC0;
    const long fbytes = (long)a.B * foc * a.Lout * 4;
    const bool fast = a.act == SEGAN_ACT_NONE && m0 + MB <= a.Rvalid &&
                      (m0 + MB <= a.OC0 || m0 >= a.OC0) && fdst != nullptr && fbytes < 0x7fffffffL;
    if (fast) {
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
          if (a.bias && add_bias) bs = epi_bias(a.bias, m0 + 32 * wm * NI, rl, h);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][e] + bs), ors,
                                                  ovo[j], rl * rowstep, 0);
        }
      }
      return;
    }
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
          if (a.act == SEGAN_ACT_TANH) v = tanhf(v);
          *o = v;
        }
      }
    }
  } else {
    constexpr bool QUAD = (S == 4 && WM == 1 && NI == 4);
    const long obytes = (long)a.B * a.Nout * a.Lout * 4;
    if (QUAD && a.act == SEGAN_ACT_NONE && a.o_padL == 0 && a.o_roll == 0 &&
        a.halo == nullptr && n0 + NPT <= a.Nout && a.Lout == 4 * a.Tcols && obytes < 0x7fffffffL) {
      // deconv forward: a lane's four phase accumulators are four consecutive output samples
      const __amdgpu_buffer_rsrc_t ors =
          __builtin_amdgcn_make_buffer_rsrc(a.out0, 0, (int)obytes, 0x00020000);
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
        if (a.bias && add_bias) bs = epi_bias(a.bias, n0, nl, h);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const u32x4 o = {__builtin_bit_cast(unsigned, acc[0][j][e] + bs),
                           __builtin_bit_cast(unsigned, acc[1][j][e] + bs),
                           __builtin_bit_cast(unsigned, acc[2][j][e] + bs),
                           __builtin_bit_cast(unsigned, acc[3][j][e] + bs)};
          __builtin_amdgcn_raw_buffer_store_b128(o, ors, ovo[j] + nl * rowstep, 0, 0);   // (*)
        }
      }
      return;
    }
    const bool qfast = obytes < 0x7fffffffL;     // 32-bit byte offsets reach every output element
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        a.out0, 0, (int)(qfast ? obytes : 0), 0x00020000);
    int cb[NJ];          // columns still to be stored by the generic form below (-1: done / masked)
#pragma unroll
    for (int j = 0; j < NJ; ++j) cb[j] = col_b[j];
    if (QUAD && qfast && a.act == SEGAN_ACT_NONE && n0 + NPT <= a.Nout) {
      // conv data gradient (HI store with a left pad, a halo and possibly a roll): what depends on
      // the COLUMN only — sample, first of the lane's four output positions, the roll's wrap — is
      // worked out once per column; interior columns store one 16-byte vector per channel row (row
      // offset in the vector offset: (*) in segan_conv_shared.h), the few columns at a row's ends and on the wrap point go
      // through the generic form below.  Measured with the stores compiled out (round 4): the
      // generic form alone was 40 % of this kernel's time on enc1 / enc2's data gradients.
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
        if (a.bias && add_bias) bs = epi_bias(a.bias, n0, nl, h);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const u32x4 o = {__builtin_bit_cast(unsigned, acc[0 % NI][j][e] + bs),
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
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (cb[j] < 0) continue;
        const int q = col_t[j];
        if (QUAD) {
          const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (n >= a.Nout) continue;
          const float bs = (a.bias && add_bias) ? a.bias[n] : 0.0f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[r][j][e] + bs;
            if (a.act == SEGAN_ACT_TANH) v[r] = tanhf(v[r]);
          }
          const size_t rowoff = (size_t)col_b[j] * a.Nout + n;
          const int i0 = 4 * q - a.o_padL;
          if (i0 >= 0 && i0 + 3 < a.Lout) {
            // interior of the row (all but ~8 of its positions): the four phases are four
            // consecutive samples, also after the roll unless they straddle its wrap point
            int ib = i0 - a.o_roll;
            if (ib < 0) ib += a.Lout;
            if (ib >= a.Lout) ib -= a.Lout;
            if (ib + 3 < a.Lout) {
              const u32x4 o = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                               __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
              if (qfast)
                __builtin_amdgcn_raw_buffer_store_b128(o, qrs, (int)((rowoff * a.Lout + ib) * 4), 0, 0);
              else
                *reinterpret_cast<u32x4*>(a.out0 + rowoff * (size_t)a.Lout + ib) = o;
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
              float* o = a.out0 + rowoff * (size_t)a.Lout + ii;
              *o = v[r];
            } else if (a.halo != nullptr) {
              const int hl = a.o_padL + a.o_padR;
              float* o = nullptr;
              if (ii < 0) o = a.halo + rowoff * hl + P;
              else if (ii - a.Lout < a.o_padR) o = a.halo + rowoff * hl + a.o_padL + (ii - a.Lout);
              if (o) { *o = v[r]; }
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
              float* o = a.out0 + rowoff * (size_t)a.Lout + ii;
              *o = v;
            } else if (a.halo != nullptr) {
              const int hl = a.o_padL + a.o_padR;
              float* o = nullptr;
              if (ii < 0) o = a.halo + rowoff * hl + P;
              else if (ii - a.Lout < a.o_padR) o = a.halo + rowoff * hl + a.o_padL + (ii - a.Lout);
              if (o) { *o = v; }
            }
          }
        }
      }
    }
  }
}

// accumulator slab of a stream-K piece: [NI*NJ*4][256] float4, thread-minor
template <int NI, int NJ>
__device__ __forceinline__ void bf2_slab_store(float* slab, const f32x16 (&acc)[NI][NJ], int tid) {
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

// ====================================================================================
// contraction
// ====================================================================================
struct Bf2Extra {
  const __bf16* wp3;    // packed weight planes
  long w_plane;         // elements between weight planes
  const __bf16* act;    // packed activation planes
  long a_plane_bytes;   // bytes between activation planes
  int ngroups;          // channel groups of 16
  int G, Qp;            // groups of 8 per sample (= 2*ngroups), positions per row
  int nld;              // DMA instructions per half of the activation tile (RLs' / 64)
};

// stages the DMA runs ahead of the MFMAs.  Measured (scripts/bench_layers.py, bf16): 2 stages ahead
// with three weight buffers (two workgroups per CU) or with half-size stages (four per CU) give the
// same total as 1 stage ahead at three per CU — the loop is not latency-bound — so the shallow form stays
#define BF2_LOOK(NPL) 1

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate): loads — LDS-DMA
// included — return in order, so "at most n outstanding" = all but the newest n have landed
__device__ __forceinline__ void bf2_wait_vm(int n) {
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0x0f71); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0f72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0f75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
    case 7: __builtin_amdgcn_s_waitcnt(0x0f77); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0f79); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0f7a); break;
    default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
  }
}

template <int MB, int NB, int WM, int U, bool OUT_HI, int SHIFTMASK, int NPL, int TU>
__global__ __launch_bounds__(256, (NPL == 1 && NB == 128) ? 3 : 2) void corr_bf2_kernel(const CorrArgs a, const Bf2Extra x) {
  constexpr int S = 32 / U;
  constexpr int WN = 4 / WM;
  constexpr int NI = MB / (32 * WM);
  constexpr int NJ = NB / (32 * WN);
  constexpr int NPT = MB / S;
  constexpr int TCH = U / TU;          // weight stages per channel group
  constexpr int NSH = SHIFTMASK ? 2 : 1;
  constexpr int KI = 3;                // activation DMA instructions per wave, plane and group: RLs' <= 6 * 64
  static_assert(MB == 128, "weight tile: 2 DMA instructions per (tap, half)");
  static_assert(!OUT_HI || WM == 1, "T form: one wave holds all phases");
  constexpr int WPIECES = NPL * TU * 2 * MB;       // per buffer
  constexpr int WINS = NPL * TU * 2 * 2;           // DMA instructions per weight stage
  constexpr int LOOK = BF2_LOOK(NPL);              // DMA look-ahead in stages (NBUF weight buffers)
  constexpr int NBUF = LOOK + 1;
  static_assert(U / TU >= LOOK, "an activation tile must outlive the look-ahead");
  static_assert(WINS % 4 == 0, "weight instructions divide among the 4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int RLs = a.RLs;                 // padded to a multiple of 64
  u32x4* Wl0 = reinterpret_cast<u32x4*>(smem_raw);       // [NBUF][NPL][TU][2][MB]
  u32x4* Il0 = Wl0 + NBUF * WPIECES;                       // [2][NPL][2][RLs]
  const int IPIECES = NPL * 2 * RLs;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x.wp3), 0, 0x7fffffff, 0x00020000);

  const int nst = x.ngroups * TCH;     // weight stages per tile
  int tileA = blockIdx.x;
  long unit = (long)blockIdx.x * a.sk_units;
  const long unit_end = min(unit + (long)a.sk_units, a.sk_total);
  const int first_sk_tile = (int)(unit / nst);
  for (;;) {
  int tile, c0, c1;
  if (tileA < a.sk_nfull) {
    tile = tileA; c0 = 0; c1 = nst;
    if (a.tile_order) {
      // each XCD (workgroup b runs on XCD b % 8, each with its own L2) a CONTIGUOUS range of the round's
      // tiles; with the rows-first numbering below that is all row tiles of a run of column tiles: the
      // workgroups resident on an XCD together share weight AND activation stages
      const int rd = tileA / (int)gridDim.x;
      const int left = a.sk_nfull - rd * (int)gridDim.x;
      tile = rd * (int)gridDim.x + xcd_remap(blockIdx.x, left < (int)gridDim.x ? left : (int)gridDim.x);
    }
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
  // tile -> (row tile, column tile): rows first (round 6) or columns first
  const int coltile = a.tile_order ? tile / a.nrt : tile % a.ncoltiles;
  const int rowtile = a.rt0 + (a.tile_order ? tile - coltile * a.nrt : tile / a.ncoltiles);
  const int m0 = rowtile * MB;
  const int n0 = rowtile * NPT;
  if (!OUT_HI) {
    if (a.out0 == nullptr && m0 + MB <= a.OC0) continue;
    if (a.out1 == nullptr && m0 >= a.OC0) continue;
  }
  const ColTile ct = make_coltile(coltile * NB, a.Tcols, NB);

  // ---- activation DMA: descriptor rebased to the tile's first sample, per-lane offsets of the
  // positions lane + 64*i (constant for the tile; the channel group goes through the scalar offset)
  const long sample_bytes = (long)x.G * x.Qp * 16;
  const long left = (long)(a.B - ct.b0) * sample_bytes;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(x.act)) + (size_t)ct.b0 * sample_bytes, 0,
      (int)(left + 2 * x.a_plane_bytes < 0x7fffffffL ? left + 2 * x.a_plane_bytes : 0x7fffffffL), 0x00020000);
  // this wave stages half g = wave & 1, position blocks (wave >> 1) + 2k (static register indices)
  const int ig = wave & 1, ib = wave >> 1;
  int avo[KI];
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int j = lane + 64 * (ib + 2 * k);
    avo[k] = (int)0x80000000u;          // out of range: the DMA writes zeros
    if (ib + 2 * k < x.nld && j < a.RLv) {
      int s, tau;
      lds_pos_decode(ct, j, a.Tcols, a.H, s, tau);
      if (ct.b0 + s < a.B) avo[k] = (int)(((long)s * x.G * x.Qp + tau) * 16);
    }
  }
  // ---- weight DMA: per-lane byte offsets of rows 64*i + lane of the tile ----
  int wvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int wr = 64 * i + lane;
    const int wgrow = OUT_HI ? (wr / NPT) * a.NP + n0 + wr % NPT : m0 + wr;
    wvo[i] = wgrow * 16;
  }

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

  // weight stage `st` -> buffer `buf`: instruction q = wave + 4*k covers (p, tu, g, i)
  auto dma_w = [&](int st, int buf) __attribute__((always_inline)) {
    u32x4* Wl = Wl0 + buf * WPIECES;
#pragma unroll
    for (int k = 0; k < WINS / 4; ++k) {
      const int q = wave + 4 * k;
      const int i = q & 1, g = (q >> 1) & 1, tu = (q >> 2) % TU, p = q / (4 * TU);
      const long soff = (long)p * x.w_plane * 2 + ((long)((st * TU + tu) * 2 + g) * a.RP) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrs, (__attribute__((address_space(3))) void*)(Wl + ((p * TU + tu) * 2 + g) * MB + 64 * i),
          16, wvo[i], (int)soff, 0, 0);
    }
  };
  // activation tile of channel group cg -> buffer ibuf: this wave's half, its position blocks
  auto dma_i = [&](int cg, int ibuf) __attribute__((always_inline)) {
    u32x4* Il = Il0 + ibuf * IPIECES;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const long soff = (long)p * x.a_plane_bytes + ((long)(2 * cg + ig) * x.Qp) * 16;
#pragma unroll
      for (int k = 0; k < KI; ++k) {
        if (ib + 2 * k < x.nld)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              ars, (__attribute__((address_space(3))) void*)(Il + (p * 2 + ig) * RLs + 64 * (ib + 2 * k)),
              16, avo[k], (int)soff, 0, 0);
      }
    }
  };

  // DMA instructions this wave issues per stage: the weights, plus the activation tile when the
  // stage opens a channel group
  int cnt_i = 0;
#pragma unroll
  for (int k = 0; k < KI; ++k) cnt_i += (ib + 2 * k < x.nld) ? NPL : 0;
  auto issue = [&](int st) __attribute__((always_inline)) {      // everything stage `st` needs
    if (st % TCH == 0 || st == c0) dma_i(st / TCH, (st / TCH) & 1);
    dma_w(st, (st - c0) % NBUF);
  };
  auto group = [&](int st) { return WINS / 4 + ((st % TCH == 0) ? cnt_i : 0); };
  issue(c0);
  if (LOOK > 1 && c0 + 1 < c1) issue(c0 + 1);
  // (the epilogue stores of the previous tile may still be in flight and complete out of order
  // with loads: a full drain here, partial waits only inside the loop)
  __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
  __syncthreads();
  for (int st = c0; st < c1; ++st) {
    const int buf = (st - c0) % NBUF;
    const int cg = st / TCH, tc = st - cg * TCH;
    const bool ahead = st + LOOK < c1;
    if (ahead) issue(st + LOOK);
    const u32x4* Wl = Wl0 + buf * WPIECES;
    const u32x4* Il = Il0 + (cg & 1) * IPIECES;
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
    // the NEXT stage's DMA has landed: only the group issued above may still be outstanding
    bf2_wait_vm((LOOK > 1 && ahead) ? group(st + LOOK) : 0);
    __syncthreads();
  }

  if (partial)
    bf2_slab_store<NI, NJ>(a.sk_ws + (size_t)(blockIdx.x * 2 + (tile - a.sk_nfull - first_sk_tile)) * (MB * NB), acc, tid);
  else
    bf2_store_tile<MB, NB, WM, U, OUT_HI>(a, acc, m0, n0, wm, h, col_b, col_t);
  }  // tile loop
}

// Stream-K second pass (as corr_fixup_kernel of the fp32 path): one workgroup per cut tile adds
// the slabs of its pieces in stage order — a fixed order, no atomics, no zero-filled outputs —
// and stores the tile.
template <int MB, int NB, int WM, int U, bool OUT_HI>
__global__ __launch_bounds__(256) void bf2_fixup_kernel(const CorrArgs a, int nst) {
  constexpr int S = 32 / U;
  constexpr int WN = 4 / WM;
  constexpr int NI = MB / (32 * WM);
  constexpr int NJ = NB / (32 * WN);
  constexpr int NPT = MB / S;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int t = blockIdx.x;
  const long u0 = (long)t * nst, u1 = u0 + nst - 1;
  const int g_first = (int)(u0 / a.sk_units), g_last = (int)(u1 / a.sk_units);
  if (g_first == g_last) return;          // held whole by one workgroup: already stored
  const int tile = a.sk_nfull + t;
  const int coltile = a.tile_order ? tile / a.nrt : tile % a.ncoltiles;
  const int rowtile = a.rt0 + (a.tile_order ? tile - coltile * a.nrt : tile / a.ncoltiles);
  const int m0 = rowtile * MB, n0 = rowtile * NPT;
  if (!OUT_HI) {
    if (a.out0 == nullptr && m0 + MB <= a.OC0) return;
    if (a.out1 == nullptr && m0 >= a.OC0) return;
  }
  const ColTile ct = make_coltile(coltile * NB, a.Tcols, NB);
  int col_b[NJ], col_t[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cl = wn * (NB / WN) + 32 * j + l31;
    const int col = ct.col0 + cl;
    if (col < a.Ctot) {
      col_b[j] = col / a.Tcols;
      col_t[j] = col - col_b[j] * a.Tcols;
    } else {
      col_b[j] = -1;
      col_t[j] = 0;
    }
  }
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  constexpr int NV = NI * NJ * 4;
  for (int g = g_first; g <= g_last; ++g) {
    const int piece = t - (int)(((long)g * a.sk_units) / nst);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(a.sk_ws + (size_t)(g * 2 + piece) * (MB * NB));
    f32x4 v[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = s4[q * 256 + tid];
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q / (NJ * 4)][(q / 4) % NJ][4 * (q % 4) + e] += v[q][e];
  }
  bf2_store_tile<MB, NB, WM, U, OUT_HI>(a, acc, m0, n0, wm, h, col_b, col_t);
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
  // batches of 8 loads in flight per thread before their LDS stores (a rolled loop makes every
  // load its own round trip: the packing ran at 1.1 - 1.3 TB/s until round 6)
  for (int e0 = tid; e0 < RT * W; e0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = e0 + 256 * i;
      const int ml = e / W, x = e - ml * W;
      const int c = x >> 5, k = x & 31;
      const int m = m0 + ml, n = n0 + c;
      v[i] = (e < RT * W && m < M && n < N && k < K) ? w[((size_t)m * N + n) * K + k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = e0 + 256 * i;
      if (e < RT * W) {
        const int ml = e / W;
        t(ml, e - ml * W) = v[i];
      }
    }
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
      split3b(t(ml, c * 32 + S * u + r), p1, p2, p3);
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
  for (int e0 = tid; e0 < 8 * 32 * 32; e0 += 8 * 256) {      // 8 loads in flight per thread
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = e0 + 256 * i;
      const int ml = e >> 10, x = e & 1023;
      const int nl = x >> 5, k = x & 31;
      const int m = 8 * hg + ml, nn = nn0 + nl;
      v[i] = (m < M && nn < N && k < K) ? w[((size_t)m * N + nn) * K + k] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = e0 + 256 * i;
      const int ml = e >> 10, x = e & 1023;
      t[ml][(x >> 5) * 33 + (x & 31)] = v[i];
    }
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
      split3b(t[q][nl * 33 + k], p1, p2, p3);
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
static inline int bf2_groups16(int Cv) { return ceil_div(Cv, 16); }

// bytes of the packed activation operand of a launch (what the caller's scratch must hold)
size_t segan_corr_bf2_scratch_bytes(int B, int Cv, int Tcols, int H, int planes) {
  return (size_t)planes * B * (2 * bf2_groups16(Cv)) * (size_t)(Tcols + H) * 16;
}

template <int S, bool IN_HI>
static int launch_pack(const PackArgs& pa, int planes, hipStream_t st) {
  const dim3 grid((unsigned)ceil_div(pa.Qp, 256), (unsigned)pa.G, (unsigned)pa.B);
  if (planes == 3) hipLaunchKernelGGL((act_pack_kernel<S, IN_HI, 3>), grid, dim3(256), 0, st, pa);
  else hipLaunchKernelGGL((act_pack_kernel<S, IN_HI, 1>), grid, dim3(256), 0, st, pa);
  return segan_check_launch("act_pack_kernel");
}

template <int NB, int WM, int U, bool IN_HI, bool OUT_HI, int SHIFTMASK, int NPL>
static int launch_bf2(CorrArgs a, Bf2Extra x, hipStream_t st) {
  constexpr int MB = 128;
  constexpr int S = 32 / U;
  constexpr int TU = (NPL == 3) ? 1 : (U >= 8 ? 4 : U);   // bf16x3: 3 planes per tap fill the LDS
  constexpr int TCH = U / TU;
  a.RLv = a.RLs;
  a.RLs = round_up(a.RLs, 64);
  x.nld = a.RLs / 64;
  if (x.nld > 6) {
    segan_set_error("corr_bf2: window of %d positions too long", a.RLv);
    return SEGAN_EUNSUPPORTED;
  }
  const size_t lds = (size_t)((BF2_LOOK(NPL) + 1) * NPL * TU * 2 * MB + 2 * NPL * 2 * a.RLs) * 16;
  auto kern = corr_bf2_kernel<MB, NB, WM, U, OUT_HI, SHIFTMASK, NPL, TU>;
  static bool attr_done[16];
  static int occ[16];
  static size_t occ_lds[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!attr_done[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[dev] = true;
  }
  const int nrowtiles = OUT_HI ? a.NP / (MB / S) : ceil_div(a.Rvalid, MB);
  a.rt0 = (!OUT_HI && a.out0 == nullptr) ? a.OC0 / MB : 0;
  const int ntiles = (nrowtiles - a.rt0) * a.ncoltiles;
  const int nst = x.ngroups * TCH;
  // Tile order (round 6 experiment, kept behind SEGAN_BF2_ORDER=1).  A bf16 tile streams 2 x 128 x K x 2
  // bytes of operands for 16x the matrix rate of an fp32 tile, and these kernels move 2.5 - 3.8 TB/s of
  // L2-miss traffic (profiles/r06_pmc_hbm_traffic_bf16.json) — is the re-fetching of the columns-first
  // tile walk, which the fp32 kernels were measured to get for free (round 5), on THEIR critical path?
  // Rows first + XCD-contiguous ranges (the ~100 workgroups resident on an XCD cover all row tiles of
  // a few column tiles and walk the contraction in step) against the columns-first walk, alternating
  // on one box (profiles/r06_bf16_tile_order_ab.json): every layer within +-1 % (the conv data
  // gradient of enc1 +6 %, dec0 +2 % slower), the step 24.04 / 24.08 -> 23.97 / 24.04 ms.  No: the
  // re-fetches are served by the Infinity Cache here too and nothing waits for them.  Default: the
  // walk of rounds 3 - 5.
  static const int order = [] { const char* e = getenv("SEGAN_BF2_ORDER"); return e ? atoi(e) : 0; }();
  a.tile_order = order ? 1 : 0;
  a.nrt = nrowtiles - a.rt0;
  a.sk_nfull = ntiles;
  a.sk_units = 0;
  a.sk_total = 0;
  unsigned grid = (unsigned)ntiles;
  const double classic_eff = (double)ntiles / (256.0 * ceil_div(ntiles, 256));
  static int sk_on = -1;
  if (sk_on < 0) {
    const char* e = getenv("SEGAN_BF2_SK");
    sk_on = e ? atoi(e) : 1;
  }
  if (sk_on && a.act == SEGAN_ACT_NONE && ntiles >= 64 && nst >= 8 && classic_eff < 0.97) {
    if (occ[dev] == 0 || occ_lds[dev] != lds) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256,
                                                       lds) != hipSuccess || nb < 1)
        nb = 1;
      occ[dev] = nb > 4 ? 4 : nb;
      occ_lds[dev] = lds;
    }
    const int G = segan_grid_slots(occ[dev]);
    const int rem = ntiles - (ntiles / G) * G;
    if (rem > 0 && a.sk_ws != nullptr && a.sk_ws_floats >= (size_t)G * 2 * MB * NB) {
      a.sk_nfull = ntiles - rem;
      a.sk_total = (long)rem * nst;
      a.sk_units = (int)((a.sk_total + G - 1) / G);
      grid = (unsigned)G;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, x);
  segan_note_corr_launch(3, grid, a, ntiles);
  if (int e = segan_check_launch("corr_bf2_kernel")) return e;
  if (a.sk_total > 0) {
    hipLaunchKernelGGL((bf2_fixup_kernel<MB, NB, WM, U, OUT_HI>), dim3(ntiles - a.sk_nfull), dim3(256),
                       0, st, a, nst);
    return segan_check_launch("bf2_fixup_kernel");
  }
  return SEGAN_OK;
}

// shared front end: geometry, packing pass, extra arguments.  Returns SEGAN_EUNSUPPORTED (the
// caller then runs the fp32 form) when the scratch is missing or too small.
// column-tile width: 128.  256-column tiles (half the weight bytes streamed per MFMA, two instead
// of three workgroups per CU) were built and measured on every SEGAN+ layer (scripts/bench_layers.py):
// within +-4 %, slower on most; dropped.
#define BF2_NB 128

template <bool IN_HI>
static int bf2_prepare(CorrArgs& a, int U, int planes, void* scratch, size_t scratch_bytes,
                       Bf2Extra& x, hipStream_t st, int NB) {
  const int S = 32 / U;
  a.ncoltiles = ceil_div(a.Ctot, NB);
  a.RLs = NB + samples_per_tile(a.Tcols, NB) * a.H;
  const size_t need = segan_corr_bf2_scratch_bytes(a.B, a.Cv, a.Tcols, a.H, planes);
  if (scratch == nullptr || scratch_bytes < need) {
    segan_set_error("corr_bf2: scratch of %zu bytes needed", need);
    return SEGAN_EUNSUPPORTED;
  }
  // what follows the packed operand holds the stream-K slabs
  const size_t pk = (need + 255) & ~(size_t)255;
  a.sk_ws = scratch_bytes > pk ? (float*)((char*)scratch + pk) : nullptr;
  a.sk_ws_floats = scratch_bytes > pk ? (scratch_bytes - pk) / sizeof(float) : 0;
  const long in_elems = (long)a.B * (a.in.C0 + a.in.C1) * a.Lin;
  const long sample_bytes = (long)2 * bf2_groups16(a.Cv) * (a.Tcols + a.H) * 16;
  if (in_elems >= (1L << 31) || sample_bytes * 8 >= (1L << 30) || need >= (size_t)0x7fff0000) {
    segan_set_error("corr_bf2: operand too large for 32-bit tile offsets");
    return SEGAN_EUNSUPPORTED;
  }
  PackArgs pa;
  pa.identity = (!a.in.scale && !a.in.shift && !a.in.slope) ? 1 : 0;
  if (int e = segan_src_defaults(&a.in, st, "corr_bf2")) return e;
  pa.in = a.in;
  pa.out = (__bf16*)scratch;
  pa.B = a.B; pa.Cv = a.Cv;
  pa.G = 2 * bf2_groups16(a.Cv);
  pa.Qp = a.Tcols + a.H;
  pa.plane_elems = (size_t)a.B * pa.G * pa.Qp * 8;
  pa.Lin = a.Lin; pa.padL = a.padL; pa.mode = a.mode; pa.roll = a.roll; pa.win_start = a.win_start;
  int e;
  if (IN_HI) e = S == 4 ? launch_pack<4, true>(pa, planes, st)
                 : S == 2 ? launch_pack<2, true>(pa, planes, st) : launch_pack<1, true>(pa, planes, st);
  else e = launch_pack<1, false>(pa, planes, st);
  if (e) return e;
  x.act = pa.out;
  x.a_plane_bytes = (long)pa.plane_elems * 2;
  x.ngroups = bf2_groups16(a.Cv);
  x.G = pa.G;
  x.Qp = pa.Qp;
  x.nld = 0;
  return SEGAN_OK;
}

static inline int bf_f_pitch2(int M) { return round_up(M, 128); }

int segan_corr_bf2_f(CorrArgs& a, int U, const void* wp3, int planes, void* scratch,
                     size_t scratch_bytes, hipStream_t st) {
  if (a.Rvalid <= 64 || (U != 8 && U != 16)) {
    segan_set_error("corr_bf2: geometry stays on the other kernels");
    return SEGAN_EUNSUPPORTED;
  }
  Bf2Extra x;
  constexpr int NB = BF2_NB;
  if (int e = bf2_prepare<true>(a, U, planes, scratch, scratch_bytes, x, st, NB)) return e;
  x.wp3 = (const __bf16*)wp3;
  a.RP = bf_f_pitch2(a.Rvalid);
  x.w_plane = (long)x.ngroups * U * 2 * a.RP * 8;
  if (U == 8) return planes == 3 ? launch_bf2<128, 2, 8, true, false, 0, 3>(a, x, st)
                                 : launch_bf2<128, 2, 8, true, false, 0, 1>(a, x, st);
  return planes == 3 ? launch_bf2<128, 2, 16, true, false, 0, 3>(a, x, st)
                     : launch_bf2<128, 2, 16, true, false, 0, 1>(a, x, st);
}

int segan_corr_bf2_t(CorrArgs& a, int U, const void* wp3, int planes, void* scratch,
                     size_t scratch_bytes, hipStream_t st) {
  const int mask = (a.rowshift[0] ? 1 : 0) | (a.rowshift[1] ? 2 : 0) | (a.rowshift[2] ? 4 : 0) |
                   (a.rowshift[3] ? 8 : 0);
  const bool ok = (U == 8 && (mask == 8 || mask == 0)) || (U == 16 && mask == 0);
  if (!ok) {
    segan_set_error("corr_bf2: unsupported T-form geometry (U=%d, shift mask %d)", U, mask);
    return SEGAN_EUNSUPPORTED;
  }
  Bf2Extra x;
  constexpr int NB = BF2_NB;
  if (int e = bf2_prepare<false>(a, U, planes, scratch, scratch_bytes, x, st, NB)) return e;
  x.wp3 = (const __bf16*)wp3;
  // a.RP = S*NP already (t_pitch)
  x.w_plane = (long)x.ngroups * U * 2 * a.RP * 8;
  if (U == 8 && mask == 8)
    return planes == 3 ? launch_bf2<128, 1, 8, false, true, 8, 3>(a, x, st)
                       : launch_bf2<128, 1, 8, false, true, 8, 1>(a, x, st);
  if (U == 8)
    return planes == 3 ? launch_bf2<128, 1, 8, false, true, 0, 3>(a, x, st)
                       : launch_bf2<128, 1, 8, false, true, 0, 1>(a, x, st);
  return planes == 3 ? launch_bf2<128, 1, 16, false, true, 0, 3>(a, x, st)
                     : launch_bf2<128, 1, 16, false, true, 0, 1>(a, x, st);
}
