// Which XCD does workgroup b of a launch run on?  (HW_REG_XCC_ID, gfx942 / gfx950.)  Launch shapes: the tiled SpMM's (one
// 1024-thread workgroup with 144 KiB of LDS per CU), the panel kernel's (256 threads, many per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(int* out) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = (int)(x & 0xF);
    lds[0] = 1.f;
  }
  // stay a while so that the launch fills the chip
  for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  int* d;
  for (int shape = 0; shape < 2; ++shape) {
    const int blocks = shape == 0 ? 256 : 4096, threads = shape == 0 ? 1024 : 256, ldsb = shape == 0 ? 144 * 1024 : 0;
    hipMalloc(&d, blocks * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), ldsb, 0, d);
    std::vector<int> h(blocks);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < blocks; ++b) ok += h[b] == b % 8;
    printf("shape %d (%d x %d threads, %d B LDS): XCC_ID == blockIdx %% 8 for %d of %d workgroups; first 24:", shape, blocks, threads, ldsb, ok, blocks);
    for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
