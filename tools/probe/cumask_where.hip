// Where do the workgroups of a launch land?  One record per workgroup: XCC id, shader engine, CU — read from the
// hardware id registers — so that tools/cumask_probe.py can check which CUs a stream made by
// hipExtStreamCreateWithCUMask really owns (the mask's bit -> (XCD, CU) mapping is the driver's, not documented here).
// Build: hipcc -O2 -shared -fPIC --offload-arch=gfx950 cumask_where.hip -o libcumask_where.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void where_k(uint32_t* __restrict__ out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // (keep the workgroup alive for a while: a grid of several workgroups per CU then spreads over every CU it may use)
  uint64_t t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < static_cast<uint64_t>(spin)) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

extern "C" int cumask_where(uint32_t* out, int blocks, int spin, void* stream) {
  where_k<<<dim3(blocks), dim3(64), 0, reinterpret_cast<hipStream_t>(stream)>>>(out, spin);
  return static_cast<int>(hipGetLastError());
}
