// bf16_tile_probe.hip — what can an LDS-fed v_mfma_f32_32x32x16_bf16 loop sustain on an MI355X,
// as a function of the PER-WAVE tile (how many operand fragments are read per MFMA) and of the
// waves resident per SIMD?  The question behind DESIGN.md 5.1: corr_bf2_kernel runs 64 x 64 wave
// tiles (1.0 ds_read_b128 per MFMA: 128 B/clk/CU of LDS reads at the full MFMA rate = the whole
// LDS bandwidth) and reaches 0.35-0.40 of the 2.5 PF peak; is a 64 x 128 (0.75 reads per MFMA) or
// 128 x 128 (0.5) wave tile worth its registers?
//
// Standalone (no torch):  hipcc --offload-arch=gfx950 -O3 -o scripts/bf16_tile_probe
// scripts/bf16_tile_probe.hip ; scripts/bf16_tile_probe > profiles/rNN_bf16_tile_probe.json
//
// One workgroup = 256 threads = 4 waves, one per SIMD, `wps` workgroups per CU.  Every wave holds
// NI x NJ accumulators (32 x 32 tiles) and per contraction step of 16 reads NI "A" and NJ "B"
// fragments with ds_read_b128 (lane-linear: conflict-free, like the packed tiles of the product
// kernels) and issues NI*NJ MFMAs, the reads one step ahead of the MFMAs.  DMA = 1 adds the LDS
// WRITE side of the real kernel: per step the workgroup's share of a (2 NI x 2 NJ waves') operand
// tile arrives by buffer_load ... lds (16 B per lane) from an L2-resident buffer, behind a
// double buffer and a barrier per STG steps.  DMA = 2 (round 4, NOT yet run: the first thing for round
// 5) feeds the A operand differently: every wave reads its NI "A" fragments of a step STRAIGHT from
// the L2-resident buffer into registers (buffer_load_dwordx4, one step ahead, no LDS write and no
// LDS read for A) and only the B fragments travel by LDS-DMA — the question behind DESIGN.md 5.1's
// last paragraph: is an LDS-DMA instruction's 60-185 cycles of issue port the thing to avoid?
// Data are random bf16 (DVFS: zeros would give the clock back).  TF/s = MFMAs x 2*32*32*16 / wall time (hipEvents, best of 5); the shader clock is
// measured inside the kernel (s_memtime over s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct Clk {
  unsigned long long cyc, real;
};

#define STG 4     // contraction steps (of 16) per staged buffer

// LDS per workgroup: 2 buffers x STG steps x (2*NI + 2*NJ) fragments-rows of 32 x 16 bf16 = 1 KiB
template <int NI, int NJ>
struct Geo {
  static constexpr int FR = 2 * NI + 2 * NJ;          // 1 KiB fragments per step and workgroup
  static constexpr int STEP_U4 = FR * 64;             // u32x4 elements per step
  static constexpr int BUF_U4 = STG * STEP_U4;
  static constexpr int LDS_BYTES = 2 * BUF_U4 * 16;
};

template <int NI, int NJ, int WPS, int DMA>
__global__ __launch_bounds__(256, WPS) void probe(const u32x4* __restrict__ src, float* out,
                                                  Clk* clk, int nstage, int src_u4) {
  using G = Geo<NI, NJ>;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 2 * G::BUF_U4; i += 256) lds[i] = src[(i + 64 * blockIdx.x) % src_u4];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(src), 0, src_u4 * 16, 0x00020000);

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // DMA: the workgroup's fragments of one buffer, instruction q = wave + 4k moves fragment q
  constexpr int NFR = DMA == 2 ? 2 * NJ : G::FR;          // fragments per step that go through LDS-DMA
  constexpr int NDMA = (STG * NFR + 3) / 4;
  auto issue = [&](int stage, int buf) {
    if (!DMA) return;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      const int q = DMA == 2 ? (wave + 4 * k) / NFR * G::FR + 2 * NI + (wave + 4 * k) % NFR : wave + 4 * k;
      if (wave + 4 * k < STG * NFR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs, (__attribute__((address_space(3))) void*)(lds + buf * G::BUF_U4 + q * 64), 16,
            lane * 16, (int)(((unsigned)(stage * STG * G::FR + q + 7 * blockIdx.x) * 1024u) %
                             (unsigned)(src_u4 * 16 - 1024)) & ~1023, 0, 0);
    }
  };

  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  issue(0, 1);
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    const u32x4* L = lds + buf * G::BUF_U4;
    bf16x8 a0[NI], b0[NJ], a1[NI], b1[NJ];
    auto rd = [&](int u, bf16x8 (&a)[NI], bf16x8 (&b)[NJ]) {
      const u32x4* S = L + u * G::STEP_U4 + lane;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (DMA == 2) {       // this wave's own fragment, 1 KiB straight from L2 into registers
          const unsigned off = (((unsigned)((st * STG + u) * 2 * NI + wm * NI + i) + 11u * blockIdx.x) * 1024u) %
                               (unsigned)(src_u4 * 16 - 1024);
          a[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)(off & ~1023u), 0));
        } else {
          a[i] = __builtin_bit_cast(bf16x8, S[(wm * NI + i) * 64]);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = __builtin_bit_cast(bf16x8, S[(2 * NI + wn * NJ + j) * 64]);
    };
    auto mm = [&](const bf16x8 (&a)[NI], const bf16x8 (&b)[NJ]) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    rd(0, a0, b0);
#pragma unroll
    for (int u = 0; u < STG; u += 2) {
      rd(u + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mm(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < STG) rd(u + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mm(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DMA) {
      __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the other buffer has landed
      __syncthreads();
      if (st + 2 < nstage + 1) issue(st + 2, buf);      // refill the buffer just consumed
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) {
    clk[blockIdx.x].cyc = c1 - c0;
    clk[blockIdx.x].real = r1 - r0;
  }
}

template <int NI, int NJ, int WPS, int DMA>
static void run(const u32x4* src, int src_u4, float* out, Clk* clk, int ncu, bool& first) {
  using G = Geo<NI, NJ>;
  auto kern = probe<NI, NJ, WPS, DMA>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int occ = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256,
                                                     G::LDS_BYTES));
  if (occ < WPS) {
    printf("%s  {\"wave_tile\": \"%dx%d\", \"waves_per_simd_asked\": %d, \"dma\": %d, \"skipped\": "
           "\"only %d workgroups fit a CU (LDS %d B)\"}", first ? "" : ",\n", 32 * NI, 32 * NJ, WPS, DMA,
           occ, G::LDS_BYTES);
    first = false;
    return;
  }
  const int grid = ncu * WPS;
  const int nstage = 24000 / (NI * NJ * STG) / WPS + 2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, 0, src, out, clk, nstage, src_u4);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<Clk> h(grid);
  CHECK(hipMemcpy(h.data(), clk, grid * sizeof(Clk), hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int i = 0; i < grid; ++i) { cyc += (double)h[i].cyc; real += (double)h[i].real; }
  const double mhz = cyc / real * 100.0;
  const double nmfma = (double)grid * 4 * nstage * STG * NI * NJ;
  const double tf = nmfma * 2.0 * 32 * 32 * 16 / (best * 1e-3) / 1e12;
  const double cyc_per_mfma = (cyc / grid) / ((double)nstage * STG * NI * NJ * WPS);
  const double reads_per_mfma = (double)(NI + NJ) / (NI * NJ);
  printf("%s  {\"wave_tile\": \"%dx%d\", \"ds_read_b128_per_mfma\": %.2f, \"waves_per_simd\": %d, "
         "\"dma\": %d, \"tflops\": %.0f, \"frac_of_2500\": %.3f, \"shader_mhz\": %.0f, "
         "\"simd_cycles_per_mfma\": %.1f, \"lds_read_bytes_per_clk_cu\": %.0f, \"ms\": %.3f, "
         "\"lds_bytes_per_wg\": %d}",
         first ? "" : ",\n", 32 * NI, 32 * NJ, reads_per_mfma, WPS, DMA, tf, tf / 2500.0, mhz,
         cyc_per_mfma, 4.0 * reads_per_mfma * 1024.0 / cyc_per_mfma, best, G::LDS_BYTES);
  first = false;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int src_u4 = 1 << 18;       // 4 MiB of operand bytes: L2-resident
  u32x4* src;
  float* out;
  Clk* clk;
  CHECK(hipMalloc(&src, (size_t)src_u4 * 16));
  CHECK(hipMalloc(&out, (size_t)ncu * 4 * 256 * sizeof(float)));
  CHECK(hipMalloc(&clk, (size_t)ncu * 4 * sizeof(Clk)));
  // random bf16 in (-1, 1): sign | exponent 0x3f00..0x3f7f region | mantissa
  std::vector<unsigned short> h((size_t)src_u4 * 8);
  srand(1234);
  for (auto& v : h) v = (unsigned short)(((rand() & 1) << 15) | (0x3e00 + (rand() & 0x1ff)));
  CHECK(hipMemcpy(src, h.data(), (size_t)src_u4 * 16, hipMemcpyHostToDevice));
  printf("{\"device\": \"%s\", \"cus\": %d, \"instruction\": \"v_mfma_f32_32x32x16_bf16\", "
         "\"dense_peak_tflops\": 2500, \"rows\": [\n", prop.gcnArchName, ncu);
  bool first = true;
  // 64 x 64 (the product kernels), 64 x 128, 128 x 128 per wave; without and with the DMA stream
  run<2, 2, 1, 0>(src, src_u4, out, clk, ncu, first);
  run<2, 2, 2, 0>(src, src_u4, out, clk, ncu, first);
  run<2, 2, 3, 0>(src, src_u4, out, clk, ncu, first);
  run<2, 2, 2, 1>(src, src_u4, out, clk, ncu, first);
  run<2, 2, 3, 1>(src, src_u4, out, clk, ncu, first);
  run<2, 4, 1, 0>(src, src_u4, out, clk, ncu, first);
  run<2, 4, 2, 0>(src, src_u4, out, clk, ncu, first);
  run<2, 4, 1, 1>(src, src_u4, out, clk, ncu, first);
  run<2, 4, 2, 1>(src, src_u4, out, clk, ncu, first);
  run<4, 4, 1, 0>(src, src_u4, out, clk, ncu, first);
  run<4, 4, 1, 1>(src, src_u4, out, clk, ncu, first);
  // A straight from L2 into registers, B by LDS-DMA
  run<2, 2, 2, 2>(src, src_u4, out, clk, ncu, first);
  run<2, 4, 1, 2>(src, src_u4, out, clk, ncu, first);
  run<4, 2, 1, 2>(src, src_u4, out, clk, ncu, first);
  run<4, 4, 1, 2>(src, src_u4, out, clk, ncu, first);
  printf("\n]}\n");
  return 0;
}
