// Shared device/host helpers of libsegan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/segan_hip.h"

#define SEGAN_OK 0
#define SEGAN_EINVAL (-1)
#define SEGAN_ELAUNCH (-2)
#define SEGAN_EUNSUPPORTED (-3)

// error string shared by all translation units (defined in segan_api.hip)
void segan_set_error(const char* fmt, ...);
int segan_check_launch(const char* what);

#define SEGAN_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      segan_set_error(__VA_ARGS__);              \
      return SEGAN_EINVAL;                       \
    }                                            \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Blocked accumulation of the fp32 contractions.  One MFMA accumulator summed over a whole
// contraction (up to 63 488 terms: 31 744 sequential roundings) carries a forward error of
// ~4e-6 against fp64, four times what a blocked CPU sum leaves.  The big kernels therefore
// run the matrix cores into `acc` for SEGAN_ACC_BLOCK chunks (8 x 32 = 256 terms), add that
// block to a second register set and restart `acc` from zero: the rounding error grows with
// sqrt(256/2) + sqrt(K/256) instead of sqrt(K/2).  64 VALU adds + 64 moves per block of 512
// MFMAs per wave; the second set fits the 256-register budget of 2 waves per SIMD.
#define SEGAN_ACC_BLOCK 8

template <int NI, int NJ>
__device__ __forceinline__ void acc_block_flush(f32x16 (&acc)[NI][NJ], f32x16 (&sum)[NI][NJ]) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      sum[i][j] += acc[i][j];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }
}

// Workgroup slots the grid planners leave free (segan_set_reserved_slots / SEGAN_RESERVED_SLOTS).
// The contraction kernels run on PERSISTENT grids of one workgroup per resident slot (256 CUs x
// occupancy) with equal shares of the work: a foreign kernel that holds n slots while such a grid
// is launched (RCCL's channels, one 256-thread workgroup each, during a data-parallel step) makes n
// of its workgroups wait for the others to finish — the launch then takes one more share's time,
// +25..33 %, not n / slots.  With a reserve of n the grids are planned for slots - n workgroups.
int segan_reserved_slots_value(void);
static inline int segan_grid_slots(int occ) {
  const int g = 256 * occ - segan_reserved_slots_value();
  return g < 64 ? 64 : g;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Padded coordinate p of the high-rate signal -> stored index, or -1 for an implicit
// zero.  Reflect: F.pad(..., mode='reflect') of modules.py:98 (mirror without
// repeating the edge sample); roll: the conv sees torch.roll(h, roll) =
// discriminator.py:160-172.  Mirrors segan_pytorch_amd/layout.py:hi_index.
__host__ __device__ static inline int segan_hi_index(int p, int L, int padL, int mode, int roll) {
  int i = p - padL;
  if (mode == SEGAN_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
  }
  if (i < 0 || i >= L) return -1;
  if (roll != 0) {
    i -= roll;
    if (i < 0) i += L;
    if (i >= L) i -= L;
  }
  return i;
}

// transform-on-load of a segan_src channel.  The launchers replace NULL vectors by
// device-resident ones / zeros (segan_src_defaults), so the kernels load the three
// per-channel scalars unconditionally: no pointer tests, no branches in the staging code.
#define SEGAN_MAX_XF_CH 16384

struct ChanXf {
  float sc, sh, sl;
};

__device__ __forceinline__ ChanXf segan_chan_xf(const segan_src& s, int c) {
  ChanXf x;
  x.sc = s.scale[c];
  x.sh = s.shift[c];
  x.sl = s.slope[c];
  return x;
}

__device__ __forceinline__ float segan_apply_xf(const ChanXf& x, float v) {
  v = fmaf(v, x.sc, x.sh);
  return v > 0.0f ? v : v * x.sl;
}

// fills the NULL transform vectors of `s` with defaults; returns nonzero on error
int segan_src_defaults(segan_src* s, hipStream_t st, const char* what);

// row base pointer of logical channel n of sample b
__device__ __forceinline__ const float* segan_src_row(const segan_src& s, int b, int n, int L) {
  if (n < s.C0) return s.p0 + ((size_t)b * s.C0 + n) * (size_t)L;
  return s.p1 + ((size_t)b * s.C1 + (n - s.C0)) * (size_t)L;
}

__device__ __forceinline__ float warp_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
