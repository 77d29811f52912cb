// clock_probe.hip — effective shader clock WHILE another kernel runs.
// One 64-thread workgroup samples (s_memtime = shader cycles, s_memrealtime = 100 MHz ticks)
// every `gap` realtime ticks; launched on a side stream next to the kernel under test it
// occupies one wave slot of one CU.  Built by scripts/clock_under_load.py:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/libclockprobe.so scripts/clock_probe.hip
#include <hip/hip_runtime.h>

__global__ void clock_probe_kernel(unsigned long long* buf, int nsamples, int gap) {
  if (threadIdx.x != 0) return;
  unsigned long long next = wall_clock64();
  for (int i = 0; i < nsamples; ++i) {
    unsigned long long r;
    do {
      __builtin_amdgcn_s_sleep(8);
      r = wall_clock64();
    } while (r < next);
    buf[2 * i] = __builtin_readcyclecounter();
    buf[2 * i + 1] = r;
    next = r + gap;
  }
}

extern "C" int clock_probe_launch(void* buf, int nsamples, int gap, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (unsigned long long*)buf, nsamples, gap);
  return (int)hipGetLastError();
}
