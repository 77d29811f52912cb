// mfma_issue_cost.hip — what does one extra instruction next to the fp32 MFMA stream cost?
//
// The same wave-per-SIMD harness as mfma_peak.hip, but each MFMA of the stream is accompanied by
// NV independent VALU fmas, NS SALU adds, NL ds_read_b32 (results never consumed) and NW
// ds_write_b32, all as `asm volatile` so the stream is issued exactly as written.  Reported:
// SIMD cycles per MFMA at `wps` waves per SIMD -> (cycles - 64) / extras = cost per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/mfma_issue_cost scripts/mfma_issue_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NV, int NS, int NL, int NW, int WIDE>
__global__ __launch_bounds__(256, 4) void stream(const float* __restrict__ src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = src[i & 4095];
  __syncthreads();
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 4 + i) & 4095]; b[i] = src[(tid * 4 + i + 2048) & 4095]; }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
  float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3];
  const float c = 1.0000001f, d = b[0];
  int sc = iters;
  unsigned laddr = (unsigned)(size_t)(lds) + (tid & 63) * (WIDE ? 16 : 4) + (tid >> 6) * 1024;
  float t0, t1, t2, t3;
  unsigned goff = (tid * 16) & 16383;
  unsigned laddr16 = (unsigned)(size_t)(lds) + tid * 16;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 w0 = {d, d, d, d};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(c), "v"(d));
        if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(c), "v"(d));
        if (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(c), "v"(d));
        if (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(c), "v"(d));
        if (NV >= 6) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(c), "v"(d)); }
#pragma unroll
        for (int q = 0; q < NS; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
        if (!WIDE) {
          if (NL >= 1) asm volatile("ds_read_b32 %0, %1" : "=v"(t0) : "v"(laddr));
          if (NL >= 2) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(t1) : "v"(laddr));
          if (NL >= 3) asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(t2) : "v"(laddr));
          if (NL >= 4) asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(t3) : "v"(laddr));
        } else {
          if (NL >= 1) asm volatile("ds_read_b128 %0, %1" : "=v"(w0) : "v"(laddr));
        }
        if (NW >= 1) asm volatile("ds_write_b32 %0, %1 offset:16384" :: "v"(laddr), "v"(d));
        if (WIDE == 2 && (i & 3) == 0) asm volatile("global_load_dword %0, %1, %2" : "=v"(t3) : "v"(goff), "s"(src));
        if (WIDE == 3 && (i & 3) == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w0) : "v"(goff), "s"(src));
        if (WIDE == 4 && (i & 3) == 0) asm volatile("ds_write_b128 %0, %1 offset:16384" :: "v"(laddr16), "v"(w0));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + i) & 3], b[(s + 2 * i) & 3], acc[i], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  float s = x0 + x1 + x2 + x3 + (float)sc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NV, int NS, int NL, int NW, int WIDE>
static void run(const float* src, float* out, int wps, int ncu) {
  const int iters = 1000 * 4 / wps;
  const int grid = ncu * wps;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream<NV, NS, NL, NW, WIDE>), dim3(grid), dim3(256), 0, 0, src, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double nmfma_per_simd = (double)iters * 32 * wps;
  const double tf = (double)grid * 4 * iters * 32 * 4096.0 / (best * 1e-3) / 1e12;
  const double cyc = best * 1e-3 * 2.39e9 / nmfma_per_simd;   // at the measured ~2.39 GHz
  const int extras = NV + NS + NL + NW;
  printf("  {\"wps\": %d, \"valu\": %d, \"salu\": %d, \"lds_read\": %d, \"lds_write\": %d, \"wide\": %d, \"tflops\": %.1f, "
         "\"cycles_per_mfma\": %.1f, \"extra_cycles_per_instr\": %.2f},\n", wps, NV, NS, NL, NW, WIDE, tf, cyc,
         extras ? (cyc - 64.0) / extras : 0.0);
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *src, *out;
  CHECK(hipMalloc(&src, 8192 * 4)); CHECK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4));
  std::vector<float> h(4096); srand(7);
  for (auto& v : h) v = (float)(rand() % 2001 - 1000) / 1000.0f;
  CHECK(hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  printf("{\"rows\": [\n");
  for (int wps = 1; wps <= 4; ++wps) {
    if (wps == 3) continue;
    run<0, 0, 0, 0, 0>(src, out, wps, ncu);
    run<1, 0, 0, 0, 0>(src, out, wps, ncu);
    run<2, 0, 0, 0, 0>(src, out, wps, ncu);
    run<4, 0, 0, 0, 0>(src, out, wps, ncu);
    run<6, 0, 0, 0, 0>(src, out, wps, ncu);
    run<0, 1, 0, 0, 0>(src, out, wps, ncu);
    run<0, 3, 0, 0, 0>(src, out, wps, ncu);
    run<0, 6, 0, 0, 0>(src, out, wps, ncu);
    run<0, 0, 1, 0, 0>(src, out, wps, ncu);
    run<0, 0, 2, 0, 0>(src, out, wps, ncu);
    run<0, 0, 4, 0, 0>(src, out, wps, ncu);
    run<0, 0, 1, 0, 1>(src, out, wps, ncu);
    run<0, 0, 0, 1, 0>(src, out, wps, ncu);
    run<2, 3, 1, 0, 0>(src, out, wps, ncu);
    run<1, 1, 1, 0, 0>(src, out, wps, ncu);
    run<0, 0, 0, 0, 2>(src, out, wps, ncu);   // + one global_load_dword per 4 MFMAs
    run<0, 0, 0, 0, 3>(src, out, wps, ncu);   // + one global_load_dwordx4 per 4 MFMAs
    run<0, 0, 0, 0, 4>(src, out, wps, ncu);   // + one ds_write_b128 per 4 MFMAs
  }
  printf("  {}\n]}\n");
  return 0;
}
