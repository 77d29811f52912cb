// Declarations shared by the fp32 (segan_conv.hip) and bf16 / bf16x3 (segan_conv_bf2.hip)
// contraction kernels: column bookkeeping, launch arguments, packed-weight geometry.
#pragma once
#include "segan_common.h"
#include <stdlib.h>

#define KCH 64  // contraction elements per LDS chunk (= 32 MFMA k-steps of 2)

// ------------------------------------------------------------------------------------
// column bookkeeping: the GEMM column space is the flattened (sample, time) axis.  A
// tile of NB consecutive columns may cover several short samples; in LDS every sample
// segment carries its own halo of H entries, so column `cl` of local sample s sits at
// LDS position cl + s*H and tap u of it at cl + s*H + u.
// ------------------------------------------------------------------------------------
struct ColTile {
  int col0, b0, t_first, len0;
};

__device__ __forceinline__ ColTile make_coltile(int col0, int Tcols, int NBcols) {
  ColTile t;
  t.col0 = col0;
  t.b0 = col0 / Tcols;
  t.t_first = col0 - t.b0 * Tcols;
  t.len0 = min(Tcols - t.t_first, NBcols);
  return t;
}

// LDS position j -> (local sample s, window coordinate tau)
__device__ __forceinline__ void lds_pos_decode(const ColTile& ct, int j, int Tcols, int H, int& s,
                                               int& tau) {
  if (j < ct.len0 + H) {
    s = 0;
    tau = ct.t_first + j;
  } else {
    const int jj = j - (ct.len0 + H);
    const int per = Tcols + H;
    const int q = jj / per;
    s = 1 + q;
    tau = jj - q * per;
  }
}

// XCD-aware work order.  The hardware places workgroup b of a launch on XCD b % 8, each XCD has
// its own L2, and consecutive tile indices share a weight row tile: left alone, every XCD's L2
// fetches every weight tile.  xcd_remap gives XCD x the CONTIGUOUS range of the index space
// [x*n/8, (x+1)*n/8) instead (a bijection on [0, n) for any n), so a weight row tile is streamed
// by one XCD (two at a range boundary).  Placement is a speed matter only; nothing depends on it.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int x = b & 7, k = b >> 3;
  return x * q + (x < r ? x : r) + k;
}

// ====================================================================================
// corr kernel
// ====================================================================================
struct CorrArgs {
  segan_src in;
  const float* wp;  // packed weights [KtotP][RP] (zero padded: no guards on the loads)
  float* out0;
  float* out1;
  const float* bias;
  float* halo;
  int B, Cv, Ktot, RP, Rvalid;
  int Tcols, Ctot, ncoltiles;
  int Lin;                // stored row length of the input tensor
  int padL, mode, roll;   // HI input view
  int win_start, H, RLs;  // window geometry
  int rowshift[4];
  int NP, Nout;           // T form row decode: row = r*NP + n
  int OC0, OC1, Lout, act;
  int o_padL, o_roll, o_padR;  // HI store (conv dgrad: reflect halo)
  int in_identity;             // bf16 kernels: the input has no per-channel transform
  int sk_nfull;                // tiles processed whole (strided over the grid)
  int sk_units;                // stream-K part: (tile, chunk) units per workgroup
  long sk_total;               // stream-K part: total units of the remaining tiles
  int rt0;                     // first row tile (rows below it have a NULL destination)
  size_t out0_elems, out1_elems, halo_elems;   // host side: what the bf16 stream-K zeroes
  float* sk_ws;                // fp32 stream-K: accumulator slabs of the cut tiles (caller scratch)
  size_t sk_ws_floats;
  int xf_mode;                 // input transform: 0 identity, 1 scale / slope (a zero stays zero), 2 with a shift
  int acc_block;               // fp32: blocked accumulation (SEGAN_PREC_FP32_BLOCKED; segan_common.h)
  int RLv, nld;                // corr2: valid window positions (RLs is the padded row), loads per lane
  int f_pair;                  // F form, K = 31: the packing pairs the channels (f_pair() below)
  int zphase;                  // T form, K = 31: the output phase whose tap u' = 0 is the zero tap, else -1
  int tile_order, nrt;         // bf16 kernels: 1 = tiles numbered rows-first (nrt row tiles per column tile), XCD-contiguous
};

// (*) 16-byte buffer stores take their row offset in the VECTOR offset, never in an SGPR soffset.
// gfx950 hazard found in round 4: a VALU write to the data registers of a buffer_store_dwordx4 in
// the slot right after the store can reach the store — LLVM's hazard recogniser pads that case only
// when the store has NO register soffset ("this hazard only exists if the instruction is not using
// a register in the soffset field"), which does not hold on this part: with the row in an SGPR and
// no bias load between two stores, element 0 of every second vector of the conv data gradient was
// the NEXT row's value (tests/diag/diag_dgrad_epilogue.py).  One v_add per store instead.



static inline int samples_per_tile(int Tcols, int NB) {
  if (Tcols >= NB) return (Tcols % NB == 0) ? 1 : 2;
  return (NB % Tcols == 0) ? NB / Tcols : (NB + Tcols - 2) / Tcols + 1;
}

// T form: channels are padded to whole tiles (128/S channels x S phases = 128 rows)
static inline int t_np(int N, int S) { return round_up(N, 128 / S); }

// ====================================================================================
// wgrad kernels
// ====================================================================================
struct WgradArgs {
  segan_src lo;
  segan_src hi;
  float* dw;
  int B, M, N, K, Ls, Lhi;
  int Cv;                 // N*S virtual channels
  int padL, mode, roll;
  int Ctot;               // B*Ls
  int cols_per_split;
  int H, RLw;
  int ls_magic;           // ceil(65536 / Ls): x / Ls for small x when Ls < TK
  int per_magic;          // ceil(65536 / (Ls + H)): LDS position -> sample when Ls < TK
  int bf_qc;              // bf16 kernel: time chunks per sample group
  int bf_cps;             // bf16 kernel: chunks per workgroup (split of the contraction)
  int Mp;                 // its row pitch (M rounded up to 128)
  // wgrad2_kernel
  int w2_cps, w2_nch;     // chunks per contraction split, chunks in total
  int w2_cpsample;        // chunks per sample (Ls >= 32), 0 when a chunk holds whole samples
  int w2_spc;             // samples per chunk (Ls < 32)
  int w2_lsshift;         // log2(Ls) when Ls < 32, else -1
  int w2_pw, w2_nld;      // hi window positions per virtual channel, loads per lane
  float* w2_slabs;        // deterministic mode: partial tiles [split][row tile][col tile]
};

// Bias of a lane's row in the fast epilogues: row base + rl0 for half-wave 0, base + rl0 + 4 for
// half-wave 1.  Read as WAVE-UNIFORM SCALAR loads (both candidates, selected by the lane's half): a
// per-lane global load here sits BETWEEN the tile's stores, and since loads and stores retire through
// the same in-order counter every such load waited for all the stores issued before it (global_load -
// s_waitcnt vmcnt(0) - two stores, 16 x NI times per tile: that many serialized store round trips at
// every tile end).  Found in round 6; the "~60 us per whole-tile round" of DESIGN.md 5.2.
__device__ __forceinline__ float epi_bias(const float* bias, int base, int rl0, int h) {
  typedef const __attribute__((address_space(4))) float cfloat;
  cfloat* bp = (cfloat*)bias + base;
  const float lo = bp[rl0], hi = bp[rl0 + 4];
  return h ? hi : lo;
}

// packed-weight geometry (shared by the pack kernels and the launchers)
static inline int f_pitch(int M) { return M <= 64 ? 64 : round_up(M, 128); }
static inline int f_rows(int N) { return round_up(N * 32, KCH); }
static inline int t_pitch(int N, int S) { return S * t_np(N, S); }
static inline int t_rows(int M, int S) { return round_up(M * (32 / S), KCH); }

// argument checks shared by the C-ABI translation units
// F form with K = 31 taps: the 32nd (padding) row of every ODD input channel of the packed
// weights holds row 30 of its even partner, so that the contraction kernel can merge the two
// half-empty MFMA steps of a channel pair into one (corr_mma_chunk in segan_conv.hip).  Decided by
// (N, K) alone: the packing and every consumer of the packed buffer agree by construction.
static inline int f_pair(int N, int K) { return K == 31 && N > 2 && (N & 1) == 0; }
// T form with K = 31 taps packed for padding `pad_t`: output phase r reads tap S*(U-1-u') + rho(r),
// rho = (r + pad_t) % S, so u' = 0 is the padding tap k = 31 for the phase with rho = S - 1.
static inline int t_zphase(int K, int S, int pad_t) {
  return (K == 31 && S > 1) ? (((S - 1 - pad_t) % S) + S) % S : -1;
}

static inline bool stride_ok(int S) { return S == 1 || S == 2 || S == 4; }
static inline bool precision_ok(int p) { return p == 0 || p == 1 || p == 3; }
static inline int check_src(const segan_src* s, int C, const char* what) {
  SEGAN_REQUIRE(s != nullptr && s->p0 != nullptr, "%s: source is NULL", what);
  SEGAN_REQUIRE(s->C0 > 0 && s->C1 >= 0 && s->C0 + s->C1 == C,
                "%s: channel segments %d+%d != %d", what, s->C0, s->C1, C);
  SEGAN_REQUIRE(s->C1 == 0 || s->p1 != nullptr, "%s: second segment pointer is NULL", what);
  return SEGAN_OK;
}

// direct VALU kernels of the 1-2 channel edge layers (segan_conv_edge.hip); `a` is filled
// exactly as for the MFMA forms, w is the UNPACKED weight [M][N][K]
int segan_launch_tsmall(CorrArgs& a, const float* w, int K, int M, int N, int S, int pad, hipStream_t st);
int segan_launch_fsmall(CorrArgs& a, int M, int N, int S, hipStream_t st);

// diagnostics record of the last forward / data-gradient launch (segan_debug_last_corr)
void segan_note_corr_launch(int kind, unsigned grid, const CorrArgs& a, int ntiles);
// bf16 / bf16x3 forms (segan_conv_bf2.hip); `a` is filled exactly as for the fp32 kernels:
// activations pre-packed into `scratch`, both operands by LDS-DMA; SEGAN_EUNSUPPORTED when the
// scratch is missing / too small or the geometry is not covered — the caller then runs the fp32
// form (round 1's bf16 kernels, which used to take those cases, are gone)
int segan_corr_bf2_f(CorrArgs& a, int U, const void* wp3, int planes, void* scratch,
                     size_t scratch_bytes, hipStream_t st);
int segan_corr_bf2_t(CorrArgs& a, int U, const void* wp3, int planes, void* scratch,
                     size_t scratch_bytes, hipStream_t st);
size_t segan_corr_bf2_scratch_bytes(int B, int Cv, int Tcols, int H, int planes);
// wgrad on the bf16 matrix cores (segan_wgrad_bf2.hip), planes = 1 (bf16) or 3 (bf16x3): both
// operands pre-packed into `scratch`, LDS-DMA; SEGAN_EUNSUPPORTED -> the caller runs the fp32 form
int segan_wgrad_bf2(WgradArgs& a, int U, int planes, void* scratch, size_t scratch_bytes, hipStream_t st);
// diagnostics record of the last weight-gradient launch (segan_debug_last_wgrad)
void segan_note_wgrad_launch(int kind, int tiles, int nsplit, int chunks_per_split);
size_t segan_wgrad_bf2_scratch_bytes(int B, int M, int N, int Ls, int S, int planes);
