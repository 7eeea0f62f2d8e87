// Operand / result layout of v_mfma_f32_16x16x32_bf16 on gfx950, checked against the definition
//   D[i][j] = sum_k A[i][k] * B[k][j],  i, j < 16, k < 32
// with the layout the filter kernels assume: lane l = j + 16 g (j < 16, g < 4)
//   A operand: 8 bf16 = A[j][8 g .. 8 g + 7]        B operand: 8 bf16 = B[8 g .. 8 g + 7][j]
//   result:    4 f32  = D[4 g + r][j], r < 4
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/l tools/microbench/mfma_16x16x32_layout.hip && /tmp/l
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, j = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)A[j * 32 + 8 * g + e];
    b[e] = (__bf16)B[(8 * g + e) * 16 + j];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + j] = acc[r];
}

int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  for (int i = 0; i < 16 * 32; ++i) hA[i] = (float)((i * 7 + 3) % 13 - 6);
  for (int i = 0; i < 32 * 16; ++i) hB[i] = (float)((i * 5 + 1) % 11 - 5);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0.f;
      for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 16 + j];
      ref[i * 16 + j] = s;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("v_mfma_f32_16x16x32_bf16 layout check: %d mismatches of 256 (%s)\n", bad, bad ? "WRONG" : "as assumed");
  return bad != 0;
}
