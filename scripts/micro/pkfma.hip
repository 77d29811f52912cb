// micro-benchmark: issue rate of v_pk_fma_f32 against v_fmac_f32 on gfx950.  The FMAs are spelled in
// assembly: left to the compiler, the SLP vectoriser turns the "scalar" variant into v_pk_fma_f32 too.
//   mode 0: v_fmac_f32 acc, s, v          mode 1: v_pk_fma_f32 acc2, v2, v2, acc2
//   mode 2: v_pk_fma_f32 acc2, s2, v2(op_sel: low half splat), acc2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  float x[4];
  for (int i = 0; i < 4; ++i) x[i] = 1.0f + 1e-6f * (threadIdx.x + i);
  if (MODE == 0) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(x[r]));
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)threadIdx.x + i, 1.0f};
    f32x2 av = {a, b};
    f32x2 xv[4];
    for (int i = 0; i < 4; ++i) xv[i] = f32x2{x[i], x[i] * 1.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(xv[r & 3]));
          else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(av), "v"(xv[r & 3]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
int main() {
  float* o; (void)hipMalloc(&o, 256 * 4096 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wg : {256, 1024, 2048}) for (int mode = 0; mode < 3; ++mode) {
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wg), dim3(256), 0, 0, o, 0.999f, 0.001f, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wg), dim3(256), 0, 0, o, 0.999f, 0.001f, iters);
      else hipLaunchKernelGGL(k<2>, dim3(wg), dim3(256), 0, 0, o, 0.999f, 0.001f, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wg * 256 * iters * 64 * 2 * (mode ? 2 : 1);
    printf("wg %d mode %d: %.3f ms %.1f TF/s\n", wg, mode, ms, flops / ms * 1e-9);
  }
  return 0;
}
