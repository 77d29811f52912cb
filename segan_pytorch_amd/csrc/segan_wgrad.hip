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
// C ABI
// ====================================================================================
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
