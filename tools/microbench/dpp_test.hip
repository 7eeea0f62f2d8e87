#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ int wave_shr1(int x, int lane) {
  int y = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false);
  const int x15 = __builtin_amdgcn_readlane(x, 15), x31 = __builtin_amdgcn_readlane(x, 31), x47 = __builtin_amdgcn_readlane(x, 47);
  y = lane == 16 ? x15 : lane == 32 ? x31 : lane == 48 ? x47 : y;
  return y;
}
__global__ void k(int* out, int kk) {
  int lane = threadIdx.x;
  int x = lane * 10 + 1;
  out[lane] = wave_shr1(x, lane);
  out[64 + lane] = __builtin_amdgcn_readlane(x, kk - 1);
  out[128 + lane] = __shfl_up(x, 1);
}
int main() {
  int* d; hipMalloc(&d, 192 * 4);
  k<<<1, 64>>>(d, 10);
  int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d:%d/%d/%d ", i, h[i], h[128 + i], h[64 + i]);
  printf("\n");
  return 0;
}
