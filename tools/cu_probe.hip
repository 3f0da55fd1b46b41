// Which CUs does a stream run on?  One workgroup = one record {xcc id, shader engine, CU} read from the hardware-id
// registers; tools/cu_mask_probe.py launches many long-enough workgroups on a CU-masked stream and histograms them.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libcuprobe.so tools/cu_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cu_probe_kernel(uint32_t* out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup resident for a while so that the launch spreads over every CU the stream may use
  uint64_t t0 = __builtin_readcyclecounter();
  while ((int64_t)(__builtin_readcyclecounter() - t0) < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

extern "C" int cu_probe(uint32_t* out, int nblocks, int spin, hipStream_t stream) {
  hipLaunchKernelGGL(cu_probe_kernel, dim3(nblocks), dim3(256), 0, stream, out, spin);
  return (int)hipGetLastError();
}
