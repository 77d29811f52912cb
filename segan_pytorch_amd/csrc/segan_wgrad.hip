// segan_wgrad.hip — the W contraction form (both weight gradients) on the exact-fp32 MFMA:
//   dW[m,n,S*u+r] += sum_{b,t} lo[b,m,t] * HI_r[b,n,t+u]      (forms: see segan_conv.hip)
#include "segan_conv_shared.h"

// ====================================================================================
// wgrad kernel
// ====================================================================================

// x / Ls for 0 <= x < Ls + TK (Ls >= TK: one compare; else exact multiply-shift, x < 64)
template <int TK>
__device__ __forceinline__ int wg_sdiv(int x, int Ls, int magic) {
  return (Ls >= TK) ? (x >= Ls ? 1 : 0) : ((x * magic) >> 16);
}

// partial tile of one contraction split: [NI*NJ*4][256] float4, thread-minor
template <int NI, int NJ>
__device__ __forceinline__ void wgrad_slab_store(float* slab, const f32x16 (&acc)[NI][NJ], int tid) {
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

// deterministic mode: dw[m][n][k] += sum over the contraction splits, in split order
// first level of a two-level reduction (edge layers: thousands of splits of a single tile): each
// workgroup adds a GROUP of consecutive slabs, in order, into the group's first slab
template <int MB, int NBT>
__global__ __launch_bounds__(256) void wgrad_group_kernel(float* slabs, int nsplit, int gsz) {
  constexpr int NV = (MB / 64) * (NBT / 64) * 4;
  const int tid = threadIdx.x;
  const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x, ntiles = (size_t)gridDim.x * gridDim.y;
  const int z0 = blockIdx.z * gsz, z1 = min(nsplit, z0 + gsz);
  f32x4 acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int z = z0; z < z1; ++z) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(slabs + ((size_t)z * ntiles + tile) * (MB * NBT));
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] += s4[q * 256 + tid];
  }
  f32x4* d4 = reinterpret_cast<f32x4*>(slabs + ((size_t)z0 * ntiles + tile) * (MB * NBT));
#pragma unroll
  for (int q = 0; q < NV; ++q) d4[q * 256 + tid] = acc[q];
}

// WM: wave rows of the tile that wrote the slabs (2 x 2 waves, or 1 x 4 for the 32-row tiles of
// wgrad2_kernel): the slab layout is thread-minor, so the reduction must decode (row, column) of a
// thread's elements exactly as the writer did
template <int U, int MB, int NBT, int WM = 2>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a, int nsplit, int zstep = 1) {
  constexpr int S = 32 / U;
  constexpr int WN = 4 / WM;
  constexpr int CVW = NBT / U, NI = MB / (32 * WM), NJ = NBT / (32 * WN);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int cv0 = blockIdx.x * CVW, m0 = blockIdx.y * MB;
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  for (int z = 0; z < nsplit; z += zstep) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(
        a.w2_slabs + ((size_t)(z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (MB * NBT));
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
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cc = wn * (NBT / WN) + 32 * j + l31;
    const int cv = cv0 + cc / U;
    const int u = cc % U;
    const int n = cv / S, r = cv % S;
    const int k = S * u + r;
    if (cv >= a.Cv || k >= a.K) continue;
    // dw += sum.  All loads of a block before its first store (a load after a store waits for it: one
    // in-order counter — element by element these were 16 x NI x NJ dependent round trips per thread)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float old[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (MB / WM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        old[e] = m < a.M ? a.dw[((size_t)m * a.N + n) * a.K + k] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (MB / WM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < a.M) a.dw[((size_t)m * a.N + n) * a.K + k] = old[e] + acc[i][j][e];
      }
    }
  }
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

  // ---- epilogue: dw[m][n][S*u + r] += acc (atomics), or a slab for the ordered reduction ----
  if (a.w2_slabs) {
    wgrad_slab_store<NI, NJ>(a.w2_slabs + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x +
                                           blockIdx.x) * (MB * NBT), acc, tid);
    return;
  }
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

// ====================================================================================
// wgrad_edge_kernel: the same contraction for the long edge layers (1 - 2 channels on the hi side:
// the first conv of G and of D, the last deconv of G) whose lo operand is a 150 - 630 MB stream read
// exactly once — the layer is HBM-bound, and wgrad_kernel above (a 64 x 64 tile, both operands
// staged through LDS in 32-column chunks with a barrier each) ran it at 2 - 6x the stream time.
// Here a WAVE is a self-contained stream with no LDS and no barrier in its loop:
//   * the contraction index is free to be permuted as long as both operands agree: MFMA s of a
//     32-column step takes column t0 + 16h + s from half-wave h, so lane (row l31, half h) supplies
//     lo[m][t0 + 16h .. + 15].  The lo tile of a step (32*RB rows x 128 bytes) is loaded with eight
//     lanes per row — a full line per row and instruction — into the registers that are the prefetch
//     buffer, and transposed to the fragment layout through a per-WAVE LDS tile (loaded in the
//     fragment layout directly, 32 rows x 32 bytes per instruction, the L1 took ~100 cycles an
//     instruction);
//   * the hi operand of column c = (r, u) (tap k = S*u + r) at time t is x[n][S*t + k - padL]: the
//     step's window of 32*S + 32 padded samples goes through a per-WAVE LDS buffer (1 - 3 coalesced
//     loads per channel; padding, reflection, roll and the transform applied while staging) and the
//     MFMA operands are ds_read_b32 at immediate offsets S*s off one lane address;
//   * two register sets alternate: the next step's loads are issued before the current step's
//     16 x RB x NN MFMAs.
// A workgroup's four waves share a range of steps round-robin; at the end they add their tiles through LDS
// in wave order and hand out the MB x 64 tile in the slab layout of wgrad_kernel (2 x 2 waves), so the
// ordered reduction kernels (and the atomic epilogue) are shared.
// RB: 32-row blocks (1, 2: tile of 64 rows; 4: 128 rows), NN: hi channels (1 - 2).
// Preconditions (launcher): Ls % 32 == 0, N <= 2, M <= 32*RB.
// ====================================================================================
template <int U, int RB, int NN, bool LO_ID, bool HI_ID>
__global__ __launch_bounds__(256, RB >= 4 ? 1 : 2) void wgrad_edge_kernel(const WgradArgs a, int spw) {
  constexpr int S = 32 / U;
  constexpr int MB = RB <= 2 ? 64 : 128;
  constexpr int NIW = MB / 64;                 // row blocks a wave owns after the reduction
  constexpr int NBLK = RB * NN;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][NBLK][16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int k = S * (l31 % U) + l31 / U;       // tap of this lane's tile column
  const int Ls = a.Ls, Lhi = a.Lhi;
  const int spb = Ls >> 5;                     // steps per sample
  const int nsteps = a.B * spb;
  // a workgroup owns 4*spw consecutive steps, its waves take them ROUND-ROBIN: at any time the four
  // waves read adjacent 128-byte pieces of the same rows (DRAM page locality)
  constexpr int DEPTH = RB * NN >= 4 ? 1 : 2;
  const int wg_beg = min(nsteps, (int)blockIdx.z * 4 * spw), wg_end = min(nsteps, wg_beg + 4 * spw);
  const int cnt = wg_end - wg_beg > wave ? (wg_end - wg_beg - wave + 3) / 4 : 0;   // steps of this wave

  f32x16 acc[RB][NN];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // MFMA row of this lane (transform, row mask) ...
  bool rok[RB];
  ChanXf axf[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int m = 32 * i + l31;
    rok[i] = m < a.M;
    if (!LO_ID) axf[i] = segan_chan_xf(a.lo, rok[i] ? m : 0);
  }
  // ... and the rows it LOADS: instruction q of row block i covers rows 32i + 8q .. + 7, eight lanes
  // per row (16 bytes each: one full 128-byte line per row and instruction)
  const int lrow = lane >> 3, lpc = lane & 7;
  int crow[RB][4];           // element offset of the row inside its segment's sample
  bool cs1[RB][4];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int m = 32 * i + 8 * q + lrow;
      m = m < a.M ? m : 0;
      cs1[i][q] = m >= a.lo.C0;
      crow[i][q] = (cs1[i][q] ? m - a.lo.C0 : m) * Ls;
    }
  const bool rows_full = (a.M & 31) == 0;
  bool hs1[NN];
  int hn[NN];
  ChanXf bxf[NN];
#pragma unroll
  for (int j = 0; j < NN; ++j) {
    const int n = j < a.N ? j : 0;
    hs1[j] = n >= a.hi.C0;
    hn[j] = hs1[j] ? n - a.hi.C0 : n;
    if (!HI_ID) bxf[j] = segan_chan_xf(a.hi, n);
  }

  // hi window of a step: the 32*S + 32 padded samples S*t0 .. S*t0 + 32*S + 31 of each channel, staged
  // per WAVE in LDS (no workgroup barrier: a wave's DS operations execute in order).  Read directly
  // from global memory the hi operand was 16 dword loads per channel and step at tap-permuted lane
  // addresses, and the kernel's time followed their count (42 cycles each per CU); the window is 1 - 3
  // coalesced loads per channel, with the padding / reflection / roll index and the transform applied
  // to 1 - 3 samples per lane instead of 16.
  constexpr int XWIN = 32 * S + 32, XL = (XWIN + 63) / 64;
  constexpr int WLDS = NN * XL * 64 + RB * 1024;      // floats of LDS per wave: windows, lo tile
  float* win = red + wave * WLDS;
  float* atile = win + NN * XL * 64;
  struct Regs {
    f32x4 A[RB][4];
    float X[NN][XL];
    unsigned xok;           // bit e = window sample lane + 64*e is a stored one (not an implicit zero)
  };
  // issues the loads of step st (nothing here reads a loaded value: no wait is forced; no branch
  // around a load either — with loads inside the arms of a branch the compiler's wait counters merge
  // conservatively and every step ends up waiting for the loads it has just issued)
  auto load = [&](Regs& R, int st) {
    const int b = st / spb;
    const int t0 = (st - b * spb) << 5;
    const float* sb0 = a.lo.p0 + (size_t)b * a.lo.C0 * Ls + t0 + 4 * lpc;
    const float* sb1 = a.lo.p1 + (size_t)b * a.lo.C1 * Ls + t0 + 4 * lpc;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        R.A[i][q] = *reinterpret_cast<const f32x4*>((cs1[i][q] ? sb1 : sb0) + crow[i][q]);
    R.xok = 0u;
    int off[XL];
#pragma unroll
    for (int e = 0; e < XL; ++e) {
      const int pw = lane + 64 * e;
      const int idx = segan_hi_index(S * t0 + pw, Lhi, a.padL, a.mode, a.roll);
      const bool ok = pw < XWIN && idx >= 0;
      if (ok) R.xok |= 1u << e;
      off[e] = ok ? idx : 0;
    }
#pragma unroll
    for (int j = 0; j < NN; ++j) {
      const float* q = hs1[j] ? a.hi.p1 + ((size_t)b * a.hi.C1 + hn[j]) * Lhi
                              : a.hi.p0 + ((size_t)b * a.hi.C0 + hn[j]) * Lhi;
#pragma unroll
      for (int e = 0; e < XL; ++e) R.X[j][e] = q[off[e]];
    }
  };
  // transforms and masks, the window through LDS, then the step's MFMAs
  auto compute = [&](Regs& R) {
    // lo: the coalesced pieces into the wave's [32*RB rows][32 columns] tile, the 16-byte column
    // XOR-swizzled by the row (conflict-free both ways), and back as MFMA fragments: lane (row l31,
    // half h) takes columns 16h .. 16h + 15
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(atile + (32 * i + 8 * q + lrow) * 32 + 4 * (lpc ^ lrow)) = R.A[i][q];
    f32x4 Af[RB][4];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Af[i][q] = *reinterpret_cast<const f32x4*>(atile + (32 * i + l31) * 32 + 4 * ((4 * h + q) ^ (l31 & 7)));
    if (!LO_ID || !rows_full) {
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = Af[i][q][e];
            if (!LO_ID) v = segan_apply_xf(axf[i], v);
            Af[i][q][e] = rok[i] ? v : 0.0f;
          }
    }
#pragma unroll
    for (int j = 0; j < NN; ++j)
#pragma unroll
      for (int e = 0; e < XL; ++e) {
        float v = R.X[j][e];
        if (!HI_ID) v = segan_apply_xf(bxf[j], v);
        win[j * (XL * 64) + lane + 64 * e] = ((R.xok >> e) & 1u) ? v : 0.0f;
      }
    float Bv[NN][16];
#pragma unroll
    for (int j = 0; j < NN; ++j)
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) Bv[j][s2] = win[j * (XL * 64) + S * (16 * h + s2) + k];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[i][s2 >> 2][s2 & 3], Bv[j][s2],
                                                           acc[i][j], 0, 0, 0);
  };

  if (cnt > 0) {
    // the loads are unconditional (past the end: the last step again, unused) for the same reason.
    // DEPTH register sets in flight ahead of the MFMAs: a wave's loads are 64 rows x 128 bytes at a
    // 4*Ls-byte stride, and at one step ahead the stream ran at latency x bytes in flight (measured
    // 2.6 us a step on the 16-row shape = 1.6 TB/s; 5.3 us on the 64-row one = 3.1 TB/s)
    auto stp = [&](int i) { return wg_beg + 4 * min(i, cnt - 1) + wave; };
    if (DEPTH == 1) {
      Regs R0, R1;
      load(R0, stp(0));
      for (int i = 0; i < cnt; i += 2) {
        load(R1, stp(i + 1));
        compute(R0);
        load(R0, stp(i + 2));
        if (i + 1 < cnt) compute(R1);
      }
    } else {
      Regs R0, R1, R2;
      load(R0, stp(0));
      load(R1, stp(1));
      for (int i = 0; i < cnt; i += 3) {
        load(R2, stp(i + 2));
        compute(R0);
        load(R0, stp(i + 3));
        if (i + 1 < cnt) compute(R1);
        load(R1, stp(i + 4));
        if (i + 2 < cnt) compute(R2);
      }
    }
  }

  // ---- the four waves' tiles, added in wave order; wave (wm, wn) takes row blocks wm*NIW + ii,
  // column block wn of the MB x 64 tile (the 2 x 2 layout of wgrad_kernel).  red[] overlays the
  // windows: every wave is past its last step ----
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[((wave * NBLK + i * NN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 out[NIW][1];
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii) {
    const int rb = wm * NIW + ii;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float sum = 0.0f;
      if (rb < RB && wn < NN) {
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[((w * NBLK + rb * NN + wn) * 16 + e) * 64 + lane];
      }
      out[ii][0][e] = sum;
    }
  }
  if (a.w2_slabs) {
    wgrad_slab_store<NIW, 1>(a.w2_slabs + (size_t)blockIdx.z * (MB * 64), out, tid);
    return;
  }
  const int cc = wn * 32 + l31;
  const int cv = cc / U, u = cc % U;
  const int n = cv / S, r = cv % S;
  const int kk = S * u + r;
  if (cv >= a.Cv || kk >= a.K) return;
#pragma unroll
  for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = wm * (MB / 2) + 32 * ii + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (m < a.M) atomicAdd(a.dw + ((size_t)m * a.N + n) * a.K + kk, out[ii][0][e]);
    }
}

// ====================================================================================
// wgrad2_kernel: the same contraction for the big layers (128 x 128 block tiles) with a
// staging path that leaves the matrix pipe alone (cost model: head of segan_conv.hip).
//   * lo (always identity here: the launcher materialises a transformed or two-segment lo
//     once per call) goes HBM/L2 -> LDS by LDS-DMA, 4 instructions per wave and chunk; the
//     tile is [128 rows][32 columns] with the 16-byte column XOR-swizzled by (row >> 1) & 7
//     (applied on the SOURCE address — the DMA destination is lane-linear) so that the
//     ds_read_b128 fragment reads are conflict-free without row padding;
//   * hi: a block covers 4 real channels = one per wave, so scale / shift / slope are wave-
//     uniform scalars; a lane's <= 4 elements (phase r, window position p) have byte offsets
//     that depend only on the chunk's CLASS (first / middle / last chunk of a sample, or whole
//     samples when Ls <= 32): three precomputed offset sets, everything else is the SGPR
//     offset; padding zeros and the reflect mirror are inside the sets.
// Preconditions (launcher): Ls % 32 == 0 or 32 % Ls == 0; tensors below 2 GiB per workgroup
// range.  The contraction is split over blockIdx.z; partial tiles are added to dw with fp32
// atomics, or — deterministic mode — written as slabs and added in split order by
// wgrad_reduce_kernel.
// ====================================================================================
#define WG2_TK 32
#define WG2_NLD 4

// MB: rows (low-rate channels m) of the block tile.  128 for the big layers; 64 and 32 for layers
// with at most that many low-rate channels (the 16 - 64-channel layers of the 11-layer stride-2 shape
// ran the 128-row tile at 25 - 50 % row utilisation: round-5 review, weak 3).  Waves 2 x 2 (1 x 4 for
// MB = 32); the hi staging — one real channel per wave — does not depend on MB.
template <int U, bool XF, int MB = 128>
__global__ __launch_bounds__(256, 2) void wgrad2_kernel(const WgradArgs a) {
  constexpr int S = 32 / U;
  constexpr int TK = WG2_TK;
  constexpr int NBT = 128;
  constexpr int CVW = NBT / U;        // virtual channels per block = 4 real channels x S phases
  constexpr int WM = MB >= 64 ? 2 : 1, WN = 4 / WM;
  constexpr int NI = MB / (32 * WM), NJ = NBT / (32 * WN);
  constexpr int NLD = WG2_NLD;
  constexpr int NDMA = MB / 32;       // lo DMA instructions per wave and chunk (8 rows each)
  static_assert(CVW / S == 4, "one real hi channel per wave");
  static_assert(MB == 128 || MB == 64 || MB == 32, "block rows");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RLw = a.RLw;
  float* Al0 = smem;                  // [2][MB*TK]  (swizzled 16-B columns)
  float* Bl0 = Al0 + 2 * MB * TK;     // [2][CVW*RLw]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int cv0 = blockIdx.x * CVW;
  const int m0 = blockIdx.y * MB;
  const int ch_beg = blockIdx.z * a.w2_cps;
  const int ch_end = min(ch_beg + a.w2_cps, a.w2_nch);
  if (ch_beg >= ch_end) return;
  const int Ls = a.Ls;
  const int cpsm = a.w2_cpsample;     // chunks per sample (Ls >= TK) or 0 (whole samples per chunk)
  const int spc = a.w2_spc;           // samples per chunk (Ls < TK: TK / Ls, else 1)

  // first sample of this workgroup's range; descriptors are rebased to it
  const int b_first = cpsm ? ch_beg / cpsm : ch_beg * spc;
  int tq = cpsm ? ch_beg - b_first * cpsm : 0;     // chunk index inside the sample
  int brel = 0;                                     // sample of the current chunk, relative

  // ---- lo: LDS-DMA ----
  const long lo_left = (long)(a.B - b_first) * a.M * Ls * 4;
  const __amdgpu_buffer_rsrc_t lor = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.lo.p0) + (size_t)b_first * a.M * Ls, 0,
      (int)(lo_left < 0x7fffffffL ? lo_left : 0x7fffffffL), 0x00020000);
  // instruction q = wave + 4p moves rows 8q .. 8q+7: lane -> (row 8q + rr, physical column pc)
  const int rr = lane >> 3, pc = lane & 7;
  const int lc = pc ^ ((((wave & 1) << 2) + (rr >> 1)) & 7);    // logical 16-B column
  int lo_vo;
  if (cpsm) lo_vo = (rr * Ls + 4 * lc) * 4;
  else {
    const int c4 = Ls / 4;            // 16-B columns per sample
    lo_vo = (rr * Ls + (lc / c4) * a.M * Ls + 4 * (lc % c4)) * 4;
  }

  // ---- hi: this wave's real channel ----
  const int n_hi = cv0 / S + wave;
  const bool n_ok = n_hi < a.N;
  const bool hseg1 = n_ok && n_hi >= a.hi.C0;
  const int hC = hseg1 ? a.hi.C1 : a.hi.C0;
  const int hn = hseg1 ? n_hi - a.hi.C0 : n_hi;
  const long hi_left = (long)(a.B - b_first) * hC * a.Lhi * 4;
  const __amdgpu_buffer_rsrc_t hir = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(hseg1 ? a.hi.p1 : a.hi.p0) + (size_t)b_first * hC * a.Lhi, 0,
      n_ok ? (int)(hi_left < 0x7fffffffL ? hi_left : 0x7fffffffL) : 0, 0x00020000);
  float xsc = 1.0f, xsh = 0.0f, xsl = 1.0f;
  if (XF) {
    typedef const __attribute__((address_space(4))) float* cptr;
    const int cx = n_ok ? n_hi : 0;
    xsc = ((cptr)a.hi.scale)[cx];
    xsh = ((cptr)a.hi.shift)[cx];
    xsl = ((cptr)a.hi.slope)[cx];
  }
  // element e = lane + 64*i of the wave: phase r = e / RLw, window position p = e % RLw.
  // Offset sets: [0] first chunk of a sample (or whole samples), [1] middle, [2] last.
  const int nld = a.w2_nld;
  int hvo[3][NLD];
  int hl_w[NLD];
  bool hok[NLD];      // the lane owns an element of the window (lanes past S*pw stage nothing)
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = lane + 64 * i;
    const int r = e / a.w2_pw, p = e - r * a.w2_pw;
    hok[i] = i < nld && r < S;
    hl_w[i] = ((wave * S + (r < S ? r : 0)) * RLw + p);
#pragma unroll
    for (int c = 0; c < 3; ++c) hvo[c][i] = (int)0x80000000u;
    if (i < nld && r < S) {
      if (cpsm) {
        const int i0 = segan_hi_index(S * p + r, a.Lhi, a.padL, a.mode, a.roll);
        if (i0 >= 0) hvo[0][i] = i0 * 4;
        hvo[1][i] = (S * p + r) * 4;
        const int i2 = segan_hi_index(S * (Ls - TK + p) + r, a.Lhi, a.padL, a.mode, a.roll);
        if (i2 >= 0) hvo[2][i] = i2 * 4;
      } else {
        const int per = Ls + a.H;
        const int sidx = p / per, tau = p - sidx * per;
        const int i0 = segan_hi_index(S * tau + r, a.Lhi, a.padL, a.mode, a.roll);
        if (sidx < spc && i0 >= 0) hvo[0][i] = (sidx * hC * a.Lhi + i0) * 4;
      }
    }
  }

  // ---- MFMA operand offsets ----
  // lo fragment of row block i, column group j: logical 16-B column 2j + h, swizzled
  int aoff[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = wm * (MB / WM) + 32 * i + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[i][j] = row * TK + 4 * ((2 * j + h) ^ ((row >> 1) & 7));
  }
  // hi: contraction column k = 8j + 4h + i of the chunk sits at LDS position k, plus H for
  // every completed sample when a chunk holds several (Ls < 32, a power of two >= 8)
  int bpos[4][NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int cc = wn * (NBT / WN) + 32 * jj + l31;
    const int bb = (cc / U) * RLw + cc % U + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bpos[j][jj] = bb + 8 * j + (a.w2_lsshift >= 0 ? ((8 * j) >> a.w2_lsshift) * a.H : 0);
  }

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  float hreg[NLD];
  auto load_chunk = [&](int buf) {
    // lo: NDMA (4 for the 128-row tile) DMA instructions per wave
    const int t0 = tq * TK;
    const int lo_s = ((brel * a.M + m0) * Ls + t0) * 4;
    float* Al = Al0 + buf * (MB * TK);
#pragma unroll
    for (int p = 0; p < NDMA; ++p) {
      const int q = wave + 4 * p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          lor, (__attribute__((address_space(3))) void*)(Al + q * 256), 16, lo_vo,
          lo_s + q * 8 * Ls * 4, 0, 0);
    }
    // hi
    const int rowb = (brel * hC + hn) * a.Lhi * 4;
    if (cpsm == 0 || tq == 0) {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (i < nld) hreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hir, hvo[0][i], rowb, 0));
    } else if (tq == cpsm - 1) {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (i < nld) hreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hir, hvo[2][i], rowb, 0));
    } else {
      const int so = rowb + (S * t0 - a.padL - a.roll) * 4;
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (i < nld) hreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hir, hvo[1][i], so, 0));
    }
    // advance to the next chunk
    if (cpsm) { if (++tq == cpsm) { tq = 0; ++brel; } }
    else brel += spc;
  };
  auto store_chunk = [&](int buf) {
    float* Bl = Bl0 + buf * (CVW * RLw);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (i < nld) {
        float v = hreg[i];
        if (XF) {
          v = fmaf(v, xsc, xsh);
          v = v > 0.0f ? v : v * xsl;
        }
        if (hok[i]) Bl[hl_w[i]] = v;
      }
    }
  };

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = ch_beg; ch < ch_end; ++ch) {
    const int buf = (ch - ch_beg) & 1;
    if (ch + 1 < ch_end) load_chunk(buf ^ 1);
    typedef const volatile __attribute__((address_space(3))) float* ldsp;
    const float* Al = Al0 + buf * (MB * TK);
    ldsp Bl = (ldsp)(Bl0 + buf * (CVW * RLw));
    f32x4 af0[NI], af1[NI];
    float bv0[NJ], bv1[NJ];
    auto read_a = [&](int j, f32x4 (&af)[NI]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) af[i] = *reinterpret_cast<const f32x4*>(Al + aoff[i][j]);
    };
    // step s covers contraction columns k = 8*(s/4) + 4h + (s&3)
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
    for (int j = 0; j < 4; j += 2) {
      read_a(j + 1, af1);
      read_b(4 * j + 1, bv1); SB; mma(af0, 0, bv0); SB;
      read_b(4 * j + 2, bv0); SB; mma(af0, 1, bv1); SB;
      read_b(4 * j + 3, bv1); SB; mma(af0, 2, bv0); SB;
      read_b(4 * j + 4, bv0); SB; mma(af0, 3, bv1); SB;
      if (j + 2 < 4) read_a(j + 2, af0);
      read_b(4 * j + 5, bv1); SB; mma(af1, 0, bv0); SB;
      read_b(4 * j + 6, bv0); SB; mma(af1, 1, bv1); SB;
      read_b(4 * j + 7, bv1); SB; mma(af1, 2, bv0); SB;
      if (4 * j + 8 < TK / 2) read_b(4 * j + 8, bv0);
      SB; mma(af1, 3, bv1); SB;
    }
#undef SB
    if (ch + 1 < ch_end) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  if (a.w2_slabs) {
    wgrad_slab_store<NI, NJ>(a.w2_slabs + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x +
                                           blockIdx.x) * (MB * NBT), acc, tid);
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cc = wn * (NBT / WN) + 32 * j + l31;
    const int cv = cv0 + cc / U;
    const int u = cc % U;
    const int n = cv / S, r = cv % S;
    const int k = S * u + r;
    if (cv >= a.Cv || k >= a.K) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (MB / WM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < a.M) atomicAdd(a.dw + ((size_t)m * a.N + n) * a.K + k, acc[i][j][e]);
      }
  }
}

// lo with a transform (the deconv layers' input) or in two segments: materialised once per
// call as a plain [B][M][Ls] tensor (one streaming pass; the contraction re-reads it from
// every one of the N*S/16 column tiles)
__global__ void wgrad_lo_materialize_kernel(const segan_src lo, float* out, int B, int M, int Ls4,
                                            int xf) {
  const long total = (long)B * M * Ls4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int t4 = (int)(e % Ls4);
    const long bm = e / Ls4;
    const int m = (int)(bm % M);
    const int b = (int)(bm / M);
    const float* src = m < lo.C0 ? lo.p0 + ((size_t)b * lo.C0 + m) * (size_t)(4 * Ls4)
                                 : lo.p1 + ((size_t)b * lo.C1 + (m - lo.C0)) * (size_t)(4 * Ls4);
    f32x4 v = reinterpret_cast<const f32x4*>(src)[t4];
    if (xf) {
      const ChanXf x = segan_chan_xf(lo, m);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = segan_apply_xf(x, v[q]);
    }
    reinterpret_cast<f32x4*>(out)[e] = v;
  }
}

// diagnostics: how the last fp32 weight gradient of this thread was launched
static thread_local int g_last_wgrad[6];
// kind: 1 wgrad_kernel, 2 wgrad2_kernel, 3 wgrad_bf2_kernel (bf16 / bf16x3)
void segan_note_wgrad_launch(int kind, int tiles, int nsplit, int chunks_per_split) {
  g_last_wgrad[0] = kind; g_last_wgrad[1] = tiles; g_last_wgrad[2] = nsplit;
  g_last_wgrad[3] = chunks_per_split; g_last_wgrad[4] = 0; g_last_wgrad[5] = 0;
}
extern "C" void segan_debug_last_wgrad(int* out6) {
  for (int i = 0; i < 6; ++i) out6[i] = g_last_wgrad[i];
}

// ---- wgrad2 launch ----
static int wg_cur_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d & 15;
}

// scratch layout of the fp32 weight gradient: [lo materialised: B*M*Ls floats, when lo has a
// transform or two segments][slabs: blocks * 128*128 floats, deterministic mode]
static size_t wgrad2_slab_floats(int tiles, int nsplit, int MB = 128) { return (size_t)tiles * nsplit * MB * 128; }

static bool wgrad2_geometry_ok(const WgradArgs& a, int U) {
  if (a.Cv <= 64 / U) return false;                    // edge layers: small-tile kernels
  if (a.Ls >= WG2_TK) return a.Ls % WG2_TK == 0;
  return a.Ls >= 8 && (WG2_TK % a.Ls) == 0;
}

static void wgrad2_plan(const WgradArgs& a, int U, int occ, int& tiles, int& nsplit, int& cps,
                        int& nch, int MB = 128) {
  const int ncol = ceil_div(a.Cv, 128 / U), nrow = ceil_div(a.M, MB);
  tiles = ncol * nrow;
  nch = ceil_div(a.Ctot, WG2_TK);
  const int G = segan_grid_slots(occ);
  // equal work per workgroup: aim at whole rounds of resident workgroups
  int best = 1;
  double best_eff = 0.0;
  const int ns_max = nch / 8 > 0 ? nch / 8 : 1;           // at least 8 chunks per workgroup
  for (int ns = 1; ns <= ns_max && (long)tiles * ns <= 4L * G; ++ns) {
    const long blocks = (long)tiles * ns;
    const double eff = (double)blocks / ((double)G * ((blocks + G - 1) / G));
    // fewer, longer workgroups win ties (one prologue / epilogue each)
    if (eff > best_eff + 0.02) { best_eff = eff; best = ns; }
  }
  cps = ceil_div(nch, best);
  nsplit = ceil_div(nch, cps);
}

template <int U, bool XF, int MB = 128>
static int launch_wgrad2(WgradArgs& a, hipStream_t st, bool deterministic, float* slabs,
                         size_t slab_floats_avail) {
  constexpr int TK = WG2_TK;
  constexpr int S = 32 / U;
  a.H = U - 1;
  const bool multi = a.Ls < TK;
  a.w2_cpsample = multi ? 0 : a.Ls / TK;
  a.w2_spc = multi ? TK / a.Ls : 1;
  a.w2_lsshift = -1;
  if (multi) { int sh = 0; while ((1 << sh) < a.Ls) ++sh; a.w2_lsshift = sh; }
  a.w2_pw = multi ? a.w2_spc * (a.Ls + a.H) : TK + a.H;
  // row stride = 8 or 24 (mod 32): the 4 channels x 8 taps a half-wave reads hit 32 distinct
  // banks either way; the smaller one keeps 4 workgroups per CU when a chunk holds two samples
  {
    const int r8 = a.w2_pw + (8 - a.w2_pw % 32 + 32) % 32, r24 = a.w2_pw + (24 - a.w2_pw % 32 + 32) % 32;
    a.RLw = U == 8 ? (r8 < r24 ? r8 : r24) : r8;
  }
  a.w2_nld = ceil_div(S * a.w2_pw, 64);
  if (a.w2_nld > WG2_NLD) return SEGAN_EUNSUPPORTED;
  const size_t lds = (size_t)(2 * MB * TK + 2 * (128 / U) * a.RLw) * sizeof(float);
  auto kern = wgrad2_kernel<U, XF, MB>;
  static bool attr_done[16];
  static int occ_c[16];
  static size_t occ_lds[16];
  const int d = wg_cur_device();
  if (!attr_done[d]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[d] = true;
  }
  if (occ_c[d] == 0 || occ_lds[d] != lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256,
                                                     lds) != hipSuccess || nb < 1)
      nb = 2;
    occ_c[d] = nb > 4 ? 4 : nb;
    occ_lds[d] = lds;
  }
  int tiles, nsplit, cps, nch;
  wgrad2_plan(a, U, occ_c[d], tiles, nsplit, cps, nch, MB);
  a.w2_cps = cps;
  a.w2_nch = nch;
  a.w2_slabs = nullptr;
  // an unsplit contraction touches every dw element exactly once: nothing to order
  if (nsplit == 1) deterministic = false;
  if (deterministic) {
    if (slabs == nullptr || slab_floats_avail < wgrad2_slab_floats(tiles, nsplit, MB)) {
      segan_set_error("wgrad: deterministic mode needs %zu bytes of scratch for the partial tiles",
                      wgrad2_slab_floats(tiles, nsplit, MB) * sizeof(float));
      return SEGAN_EINVAL;
    }
    a.w2_slabs = slabs;
  }
  const int ncol = ceil_div(a.Cv, 128 / U), nrow = ceil_div(a.M, MB);
  g_last_wgrad[0] = 2; g_last_wgrad[1] = tiles; g_last_wgrad[2] = nsplit; g_last_wgrad[3] = cps;
  g_last_wgrad[4] = occ_c[d]; g_last_wgrad[5] = a.w2_nld;
  hipLaunchKernelGGL(kern, dim3(ncol, nrow, nsplit), dim3(256), lds, st, a);
  if (int e = segan_check_launch("wgrad2_kernel")) return e;
  if (deterministic) {
    hipLaunchKernelGGL((wgrad_reduce_kernel<U, MB, 128, (MB >= 64 ? 2 : 1)>), dim3(ncol, nrow), dim3(256), 0,
                       st, a, nsplit, 1);
    return segan_check_launch("wgrad_reduce_kernel");
  }
  return SEGAN_OK;
}

// the row tile by the number of low-rate channels: 32 rows for M <= 32, 64 for M <= 64 (stride 2
// only: the shapes that need it), 128 otherwise
template <int U, bool XF>
static int launch_wgrad2_rows(WgradArgs& a, hipStream_t st, bool deterministic, float* slabs,
                              size_t slab_floats_avail) {
  if constexpr (U == 16) {
    if (a.M <= 32) return launch_wgrad2<U, XF, 32>(a, st, deterministic, slabs, slab_floats_avail);
    if (a.M <= 64) return launch_wgrad2<U, XF, 64>(a, st, deterministic, slabs, slab_floats_avail);
  }
  return launch_wgrad2<U, XF, 128>(a, st, deterministic, slabs, slab_floats_avail);
}

template <int U, bool LO_ID, bool HI_ID, int MB, int NBT>
static int launch_wgrad_tile(WgradArgs& a, hipStream_t st, float* slabs, size_t slab_floats) {
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
  a.ls_magic = (65536 + a.Ls - 1) / a.Ls;
  a.per_magic = (65536 + a.Ls + a.H - 1) / (a.Ls + a.H);
  const int ncol = ceil_div(a.Cv, CVW);
  const int nrow = ceil_div(a.M, MB);
  // split the (b,t) contraction so the grid has a few workgroups per CU
  const int tiles = ncol * nrow;
  const int chunks = ceil_div(a.Ctot, TK);
  int nsplit = ceil_div(1536, tiles);
  if (nsplit > chunks / 4) nsplit = chunks / 4;   // at least 4 chunks per workgroup
  if (nsplit < 1) nsplit = 1;
  const int chunks_per = ceil_div(chunks, nsplit);
  nsplit = ceil_div(chunks, chunks_per);
  a.cols_per_split = chunks_per * TK;
  const size_t lds = (size_t)(2 * MB * (TK + 4) + 2 * CVW * a.RLw) * sizeof(float);
  auto kern = wgrad_kernel<U, TK, LO_ID, HI_ID, MB, NBT>;
  static bool attr_done[16];
  const int d = wg_cur_device();
  if (!attr_done[d]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[d] = true;
  }
  a.w2_slabs = nullptr;
  if (nsplit == 1) slabs = nullptr;      // unsplit: every dw element is touched exactly once
  if (slabs) {
    if (slab_floats < (size_t)tiles * nsplit * MB * NBT) {
      segan_set_error("wgrad: deterministic mode needs %zu bytes of scratch for the partial tiles",
                      (size_t)tiles * nsplit * MB * NBT * sizeof(float));
      return SEGAN_EINVAL;
    }
    a.w2_slabs = slabs;
  }
  g_last_wgrad[0] = 1; g_last_wgrad[1] = tiles; g_last_wgrad[2] = nsplit; g_last_wgrad[3] = chunks_per;
  g_last_wgrad[4] = 0; g_last_wgrad[5] = 0;
  hipLaunchKernelGGL(kern, dim3(ncol, nrow, nsplit), dim3(256), lds, st, a);
  if (int e = segan_check_launch("wgrad_kernel")) return e;
  if (slabs) {
    // few tiles, many splits (the 1-2 channel edge layers): a fixed two-level tree, so that the
    // reduction runs on many workgroups; the grouping depends only on the geometry
    int zstep = 1;
    if (nsplit >= 128 && tiles <= 8) {
      zstep = 32;
      hipLaunchKernelGGL((wgrad_group_kernel<MB, NBT>), dim3(ncol, nrow, ceil_div(nsplit, zstep)),
                         dim3(256), 0, st, slabs, nsplit, zstep);
      if (int e = segan_check_launch("wgrad_group_kernel")) return e;
    }
    hipLaunchKernelGGL((wgrad_reduce_kernel<U, MB, NBT>), dim3(ncol, nrow), dim3(256), 0, st, a, nsplit,
                       zstep);
    return segan_check_launch("wgrad_reduce_kernel");
  }
  return SEGAN_OK;
}

// ---- wgrad_edge launch ----
static bool wgrad_edge_geometry_ok(const WgradArgs& a, int U) {
  if (a.Cv > 64 / U || a.N > 2) return false;          // 1 - 2 channels on the hi side
  if (a.Ls % 32 != 0) return false;
  if (a.M > 128 || (a.M > 64 && a.N > 1)) return false;
  return true;
}

template <int U, int RB, int NN, bool LO_ID, bool HI_ID>
static int launch_wgrad_edge_tile(WgradArgs& a, hipStream_t st, float* slabs, size_t slab_floats) {
  constexpr int MB = RB <= 2 ? 64 : 128;
  if ((long)a.B * a.M * a.Ls >= (1L << 31) || (long)a.B * a.N * a.Lhi >= (1L << 31)) {
    segan_set_error("wgrad: operand exceeds the 2^31 element indexing limit");
    return SEGAN_EUNSUPPORTED;
  }
  if (int e = segan_src_defaults(&a.lo, st, "wgrad(lo)")) return e;
  if (int e = segan_src_defaults(&a.hi, st, "wgrad(hi)")) return e;
  // a wave per range of 32-column steps: two workgroups per CU, at least 8 steps per wave; the
  // decomposition depends on the geometry alone (deterministic mode)
  const int nsteps = a.B * (a.Ls / 32);
  // workgroups per CU: the register budget (__launch_bounds__ of the kernel).  Measured: 3 - 4 per CU
  // for the 32-row instances (they fit) change nothing
  constexpr int OCC = RB >= 4 ? 1 : 2;
  int nwg = ceil_div(nsteps, 4 * 8);
  if (nwg > 256 * OCC) nwg = 256 * OCC;
  const int spw = ceil_div(nsteps, 4 * nwg);
  nwg = ceil_div(nsteps, 4 * spw);
  a.w2_slabs = nullptr;
  if (nwg == 1) slabs = nullptr;         // one tile: every dw element is touched exactly once
  if (slabs) {
    if (slab_floats < (size_t)nwg * MB * 64) {
      segan_set_error("wgrad: deterministic mode needs %zu bytes of scratch for the partial tiles",
                      (size_t)nwg * MB * 64 * sizeof(float));
      return SEGAN_EINVAL;
    }
    a.w2_slabs = slabs;
  }
  g_last_wgrad[0] = 4; g_last_wgrad[1] = 1; g_last_wgrad[2] = nwg; g_last_wgrad[3] = spw;
  g_last_wgrad[4] = 0; g_last_wgrad[5] = 0;
  constexpr int XL = (32 * (32 / U) + 32 + 63) / 64;
  const size_t lds_red = (size_t)4 * RB * NN * 1024 * sizeof(float);
  const size_t lds_win = (size_t)4 * (NN * XL * 64 + RB * 1024) * sizeof(float);
  const size_t lds = lds_red > lds_win ? lds_red : lds_win;
  auto kern = wgrad_edge_kernel<U, RB, NN, LO_ID, HI_ID>;
  static bool attr_done[16];
  const int d = wg_cur_device();
  if (!attr_done[d]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[d] = true;
  }
  hipLaunchKernelGGL(kern, dim3(1, 1, nwg), dim3(256), lds, st, a, spw);
  if (int e = segan_check_launch("wgrad_edge_kernel")) return e;
  if (slabs) {
    int zstep = 1;
    if (nwg >= 128) {
      zstep = 32;
      hipLaunchKernelGGL((wgrad_group_kernel<MB, 64>), dim3(1, 1, ceil_div(nwg, zstep)), dim3(256), 0,
                         st, slabs, nwg, zstep);
      if (int e = segan_check_launch("wgrad_group_kernel")) return e;
    }
    hipLaunchKernelGGL((wgrad_reduce_kernel<U, MB, 64>), dim3(1, 1), dim3(256), 0, st, a, nwg, zstep);
    return segan_check_launch("wgrad_reduce_kernel");
  }
  return SEGAN_OK;
}

template <int U, bool LO_ID, bool HI_ID>
static int launch_wgrad_edge(WgradArgs& a, hipStream_t st, float* slabs, size_t slab_floats) {
  if (a.N == 1) {
    if (a.M <= 32) return launch_wgrad_edge_tile<U, 1, 1, LO_ID, HI_ID>(a, st, slabs, slab_floats);
    if (a.M <= 64) return launch_wgrad_edge_tile<U, 2, 1, LO_ID, HI_ID>(a, st, slabs, slab_floats);
    return launch_wgrad_edge_tile<U, 4, 1, LO_ID, HI_ID>(a, st, slabs, slab_floats);
  }
  if (a.M <= 32) return launch_wgrad_edge_tile<U, 1, 2, LO_ID, HI_ID>(a, st, slabs, slab_floats);
  return launch_wgrad_edge_tile<U, 2, 2, LO_ID, HI_ID>(a, st, slabs, slab_floats);
}

template <int U, bool LO_ID, bool HI_ID>
static int launch_wgrad_x(WgradArgs& a, hipStream_t st, float* slabs, size_t slab_floats) {
  // long edge layers: the streaming kernel
  static const bool edge_on = getenv("SEGAN_WGRAD_EDGE") == nullptr || atoi(getenv("SEGAN_WGRAD_EDGE")) != 0;
  if (edge_on && wgrad_edge_geometry_ok(a, U)) return launch_wgrad_edge<U, LO_ID, HI_ID>(a, st, slabs, slab_floats);
  // edge layers (1-2 channels on the hi side: N*S <= 64/U virtual channels): 64 columns
  // suffice, and 64 rows when M <= 64
  if (a.Cv <= 64 / U) {
    if (a.M <= 64) return launch_wgrad_tile<U, LO_ID, HI_ID, 64, 64>(a, st, slabs, slab_floats);
    return launch_wgrad_tile<U, LO_ID, HI_ID, 128, 64>(a, st, slabs, slab_floats);
  }
  return launch_wgrad_tile<U, LO_ID, HI_ID, 128, 128>(a, st, slabs, slab_floats);
}

template <int U>
static int launch_wgrad_t(WgradArgs& a, hipStream_t st, float* slabs, size_t slab_floats) {
  const bool lo_id = !a.lo.scale && !a.lo.shift && !a.lo.slope;
  const bool hi_id = !a.hi.scale && !a.hi.shift && !a.hi.slope;
  if (lo_id && hi_id) return launch_wgrad_x<U, true, true>(a, st, slabs, slab_floats);
  if (lo_id) return launch_wgrad_x<U, true, false>(a, st, slabs, slab_floats);
  if (hi_id) return launch_wgrad_x<U, false, true>(a, st, slabs, slab_floats);
  return launch_wgrad_x<U, false, false>(a, st, slabs, slab_floats);
}

// the fast kernel where its preconditions hold, the general one otherwise
template <int U>
static int launch_wgrad_fp32(WgradArgs& a, hipStream_t st, bool deterministic, float* scratch,
                             size_t scratch_floats) {
  float* slabs = nullptr;
  size_t slab_floats = 0;
  const size_t lo_floats = (size_t)a.B * a.M * a.Ls;
  const bool lo_plain = !a.lo.scale && !a.lo.shift && !a.lo.slope && a.lo.C1 == 0;
  bool fast = wgrad2_geometry_ok(a, U);
  if (fast && !lo_plain && (scratch == nullptr || scratch_floats < lo_floats)) fast = false;
  size_t used = (fast && !lo_plain) ? lo_floats : 0;
  if (deterministic) {
    if (scratch == nullptr) {
      segan_set_error("wgrad: deterministic mode needs scratch (segan_wgrad_scratch_bytes)");
      return SEGAN_EINVAL;
    }
    slabs = scratch + used;
    slab_floats = scratch_floats - used;
  }
  if (fast) {
    if (!lo_plain) {
      if (int e = segan_src_defaults(&a.lo, st, "wgrad(lo)")) return e;
      const bool xf = true;
      const long total = (long)lo_floats / 4;
      const unsigned grid = (unsigned)(total / 256 > 8192 ? 8192 : (total + 255) / 256);
      hipLaunchKernelGGL(wgrad_lo_materialize_kernel, dim3(grid), dim3(256), 0, st, a.lo, scratch,
                         a.B, a.M, a.Ls / 4, xf ? 1 : 0);
      if (int e = segan_check_launch("wgrad_lo_materialize_kernel")) return e;
      a.lo.p0 = scratch; a.lo.p1 = nullptr; a.lo.C0 = a.M; a.lo.C1 = 0;
      a.lo.scale = a.lo.shift = a.lo.slope = nullptr;
    }
    const bool hi_xf = a.hi.scale || a.hi.shift || a.hi.slope;
    if (hi_xf) {
      if (int e = segan_src_defaults(&a.hi, st, "wgrad(hi)")) return e;
    }
    const int rc = hi_xf ? launch_wgrad2_rows<U, true>(a, st, deterministic, slabs, slab_floats)
                         : launch_wgrad2_rows<U, false>(a, st, deterministic, slabs, slab_floats);
    if (rc != SEGAN_EUNSUPPORTED) return rc;
  }
  return launch_wgrad_t<U>(a, st, slabs, slab_floats);
}

// ====================================================================================
// C ABI
// ====================================================================================
extern "C" size_t segan_wgrad_scratch_bytes(int B, int M, int N, int Ls, int S, int precision,
                                            int flags) {
  if (B <= 0 || M <= 0 || N <= 0 || Ls <= 0 || !stride_ok(S)) return 0;
  if (precision != SEGAN_PREC_FP32) {
    const int planes = precision == SEGAN_PREC_BF16 ? 1 : 3;
    return segan_wgrad_bf2_scratch_bytes(B, M, N, Ls, S, planes);
  }
  // room for a materialised lo, plus (deterministic mode) the partial tiles of every split:
  // at most max(tiles, 4 rounds of 1024 resident workgroups) tiles of 128 x 128
  size_t bytes = (size_t)B * M * Ls * sizeof(float);
  if (flags & SEGAN_WGRAD_DETERMINISTIC) {
    const int U = 32 / S;
    const size_t tiles = (size_t)ceil_div(N * S, 128 / U) * ceil_div(M, 128);
    bytes += (tiles > 4096 ? tiles : 4096) * 128 * 128 * sizeof(float);
  }
  return bytes;
}

extern "C" int segan_wgrad(const segan_src* lo, const segan_src* hi, float* dw, int B, int M, int N,
                           int Ls, int K, int S, int padL, int mode, int roll, int precision,
                           int flags, void* scratch, size_t scratch_bytes, void* stream) {
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
    const int planes = precision == SEGAN_PREC_BF16 ? 1 : 3;
    WgradArgs a2 = a;
    // SEGAN_EUNSUPPORTED (geometry or scratch): the caller runs the fp32 form
    return segan_wgrad_bf2(a2, 32 / S, planes, scratch, scratch_bytes, st);
  }
  const bool det = (flags & SEGAN_WGRAD_DETERMINISTIC) != 0;
  float* sc = (float*)scratch;
  const size_t scf = scratch ? scratch_bytes / sizeof(float) : 0;
  switch (S) {
    case 4: return launch_wgrad_fp32<8>(a, st, det, sc, scf);
    case 2: return launch_wgrad_fp32<16>(a, st, det, sc, scf);
    default: return launch_wgrad_fp32<32>(a, st, det, sc, scf);
  }
}
