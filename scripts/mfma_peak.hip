// mfma_peak.hip — what does v_mfma_f32_32x32x2_f32 sustain on THIS MI355X, and at what clock?
//
// Standalone microbenchmark (no torch):  hipcc --offload-arch=gfx950 -O3 -o scripts/mfma_peak
// scripts/mfma_peak.hip ; scripts/mfma_peak > profiles/r02_mfma_peak.json
//
// Every workgroup is 256 threads = 4 waves = one wave per SIMD; `wps` workgroups per CU give
// `wps` waves per SIMD.  Each wave issues a long stream of independent fp32 MFMAs on NACC
// accumulators, cycling through 8 A and 8 B operand registers (so the operand buses toggle
// as they do in a GEMM).  Variants:
//   data   = zero | random   operands all 0.0f, or uniform(-1,1): DVFS gives the clock back
//                             on zeros (MI355X_MICROARCH.md "DVFS give-back")
//   body   = mfma            MFMAs only
//            mfma+lds        + one ds_read_b32 per MFMA (the corr/wgrad operand traffic)
//            mfma+lds+valu   + 2 VALU (fma) per MFMA (the measured 2.4 VALU per MFMA of r01)
// The effective shader clock is measured inside the kernel: s_memtime ticks (shader cycles)
// over s_memrealtime ticks (constant 100 MHz).  TF/s = 2*32*32*2 flops per MFMA / wall time
// (hipEvents around the launch, best of 5).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct Clk {
  unsigned long long cyc, real, t0, t1;
  unsigned smid, pad;
};

template <int NACC, int BODY>
__global__ __launch_bounds__(256, 4) void mfma_stream(const float* __restrict__ src, float* out,
                                                   Clk* clk, int iters) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = src[(i * 7 + blockIdx.x) & 4095];
  __syncthreads();
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(tid * 8 + i) & 4095];
    b[i] = src[(tid * 8 + i + 2048) & 4095];
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
  float v0 = a[0], v1 = b[0];
  int lofs = tid & 1023;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        float av = a[(s + i) & 7], bv = b[(s + 2 * i) & 7];
        if (BODY >= 1) {
          // operand through LDS, as the contraction kernels read theirs
          av += lds[lofs + 64 * ((s * NACC + i) & 31)];
        }
        if (BODY >= 2) {
          v0 = fmaf(v0, 1.0000001f, bv);
          v1 = fmaf(v1, 0.9999999f, av);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
      }
    }
    lofs = (lofs + 1) & 1023;
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = v0 + v1;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) {
    clk[blockIdx.x].cyc = c1 - c0;
    clk[blockIdx.x].real = r1 - r0;
    clk[blockIdx.x].t0 = r0;
    clk[blockIdx.x].t1 = r1;
    clk[blockIdx.x].smid = __smid();
  }
}

template <int NACC, int BODY>
static void run(const char* data, const float* src, float* out, Clk* clk, int wps, int ncu,
                bool first) {
  const int iters = 4000 / NACC * 4 / wps + 1;
  const int grid = ncu * wps;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_stream<NACC, BODY>), dim3(grid), dim3(256), 0, 0, src, out, clk, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<Clk> h(grid);
  CHECK(hipMemcpy(h.data(), clk, grid * sizeof(Clk), hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  // census: workgroups per CU (smid = xcc | se | cu) and how many overlapped the FIRST block's run
  int percu[1024] = {0}, maxcu = 0, ncus = 0, overlap = 0;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < grid; ++i) {
    cyc += (double)h[i].cyc; real += (double)h[i].real;
    if (percu[h[i].smid & 1023]++ == 0) ++ncus;
    if (percu[h[i].smid & 1023] > maxcu) maxcu = percu[h[i].smid & 1023];
    if (h[i].t0 < tmin) tmin = h[i].t0;
    if (h[i].t1 > tmax) tmax = h[i].t1;
  }
  for (int i = 0; i < grid; ++i) if (h[i].t0 < tmin + (h[0].t1 - h[0].t0) / 4) ++overlap;
  const double mhz = cyc / real * 100.0;               // s_memrealtime ticks at 100 MHz
  const double nmfma = (double)grid * 4 * iters * 8 * NACC;
  const double tf = nmfma * 2.0 * 32 * 32 * 2 / (best * 1e-3) / 1e12;
  // cycles one SIMD spends per MFMA it issued (wps waves share a SIMD)
  const double cyc_per_mfma = (cyc / grid) / ((double)iters * 8 * NACC * wps);
  static const char* bodies[] = {"mfma", "mfma+lds", "mfma+lds+valu"};
  printf("%s  {\"data\": \"%s\", \"body\": \"%s\", \"acc_per_wave\": %d, \"waves_per_simd\": %d, "
         "\"tflops\": %.1f, \"shader_mhz\": %.0f, \"simd_cycles_per_mfma\": %.1f, \"ms\": %.3f, \"cus_used\": %d, \"max_wg_per_cu\": %d, "
         "\"wg_started_in_first_quarter\": %d, \"wg\": %d, \"span_ms\": %.3f}",
         first ? "" : ",\n", data, bodies[BODY], NACC, wps, tf, mhz, cyc_per_mfma, best, ncus, maxcu, overlap, grid,
         (double)(tmax - tmin) / 1e5);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *src, *out;
  Clk* clk;
  CHECK(hipMalloc(&src, 4096 * sizeof(float)));
  CHECK(hipMalloc(&out, (size_t)ncu * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&clk, (size_t)ncu * 8 * sizeof(Clk)));
  std::vector<float> h(4096);
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz_max\": %d, \"instruction\": "
         "\"v_mfma_f32_32x32x2_f32\", \"peak_at_2400mhz_tflops\": 157.3, \"rows\": [\n",
         prop.gcnArchName, ncu, prop.clockRate);
  bool first = true;
  for (int d = 0; d < 2; ++d) {
    srand(1234);
    for (auto& v : h) v = d ? (float)rand() / RAND_MAX * 2.0f - 1.0f : 0.0f;
    CHECK(hipMemcpy(src, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
    const char* name = d ? "random" : "zero";
    for (int wps = 1; wps <= 4; ++wps) {
      run<4, 0>(name, src, out, clk, wps, ncu, first); first = false;
      run<4, 1>(name, src, out, clk, wps, ncu, false);
      run<4, 2>(name, src, out, clk, wps, ncu, false);
    }
    run<8, 0>(name, src, out, clk, 1, ncu, false);
    run<8, 1>(name, src, out, clk, 2, ncu, false);
  }
  printf("\n]}\n");
  return 0;
}
