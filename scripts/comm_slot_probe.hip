// comm_slot_probe.hip — a stand-in for the kernel of an RCCL collective, for measuring on ONE GPU
// what a resident collective costs the contraction kernels.
// `nchan` workgroups of 256 threads (RCCL: one workgroup per channel), 16 KB of LDS each, hold
// their slots for `ticks` of the 100 MHz wall clock while streaming through their slice of a buffer
// (read-modify-write in 16-byte pieces, like a reduce): what a link-bound ring all-reduce looks like to
// the rest of the chip — it occupies slots for a time set by the links, not by the work.
// Built by scripts/comm_overlap_probe.py:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/libcommslot.so scripts/comm_slot_probe.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void comm_slot_kernel(float* buf, size_t floats_per_wg,
                                                        unsigned long long ticks,
                                                        unsigned long long* stamps) {
  __shared__ float stage[4096];
  const unsigned long long t0 = wall_clock64();
  float4* p = reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * floats_per_wg);
  const size_t n4 = floats_per_wg / 4;
  do {
    for (size_t i = threadIdx.x; i < n4; i += 256) {
      float4 v = p[i];
      v.x = v.x * 0.5f + 1.0f; v.y = v.y * 0.5f + 1.0f; v.z = v.z * 0.5f + 1.0f; v.w = v.w * 0.5f + 1.0f;
      stage[(threadIdx.x * 4) & 4095] = v.x;
      p[i] = v;
    }
    __syncthreads();
  } while (wall_clock64() - t0 < ticks);
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = wall_clock64();
  }
}

extern "C" int comm_slot_launch(void* buf, size_t floats_per_wg, int nchan, unsigned long long ticks,
                                void* stamps, void* stream) {
  hipLaunchKernelGGL(comm_slot_kernel, dim3(nchan), dim3(256), 0, (hipStream_t)stream, (float*)buf,
                     floats_per_wg, ticks, (unsigned long long*)stamps);
  return (int)hipGetLastError();
}
