// v_mfma_scale_f32_32x32x64_f8f6f4 (MX-scaled fp8): does the per-lane K order of the operands matter?  Three K-permutations
// applied to BOTH operands give the exact product -- the instruction pairs byte t of lane half h of A with byte t of
// lane half h of B, so a kernel may load both sides in any common order ("Why not fp8").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// fp8 e4m3 (OCP) encode of small non-negative integers 0..15 exactly: value = 2^(e-7) * (1 + m/8)
__host__ __device__ inline uint8_t enc(int v) {
  if (v == 0) return 0;
  int e = 0; while ((1 << (e + 1)) <= v) ++e;     // floor(log2 v)
  int m = ((v << 3) >> e) - 8;                     // 3 mantissa bits (exact for v < 16)
  return (uint8_t)(((e + 7) << 3) | m);
}
__global__ void k(const uint8_t* A, const uint8_t* B, float* out, int variant) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  v8i a, b;
  uint8_t ab[32], bb[32];
  for (int t = 0; t < 32; ++t) {
    int kk;
    if (variant == 0) kk = 32 * h + t;                       // lane half h holds K = 32h .. 32h+31
    else if (variant == 1) kk = 16 * h + (t & 15) + 32 * (t >> 4);  // 16-K blocks interleaved between halves
    else kk = 2 * t + h;                                      // alternating
    ab[t] = A[i * 64 + kk];
    bb[t] = B[kk * 32 + i];
  }
  for (int w = 0; w < 8; ++w) {
    a[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    b[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  const int scale = 0x7F7F7F7F;  // E8M0 127 = 2^0 in every byte
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}
int main() {
  uint8_t hA[32 * 64], hB[64 * 32]; float ref[32 * 32] = {0};
  int vA[32 * 64], vB[64 * 32];
  for (int i = 0; i < 32 * 64; ++i) { vA[i] = (i * 7 + i / 64) % 9; hA[i] = enc(vA[i]); }
  for (int i = 0; i < 64 * 32; ++i) { vB[i] = (i * 5 + i / 32 * 3) % 7; hB[i] = enc(vB[i]); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 64; ++kk) s += vA[i * 64 + kk] * vB[kk * 32 + j]; ref[i * 32 + j] = s; }
  uint8_t *dA, *dB; float* dO;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dO, sizeof(ref));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int variant = 0; variant < 3; ++variant) {
    k<<<1, 64>>>(dA, dB, dO, variant);
    float out[32 * 32];
    hipMemcpy(out, dO, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0; double maxd = 0;
    for (int t = 0; t < 32 * 32; ++t) { double d = fabs(out[t] - ref[t]); if (d > 1e-3) ++bad; if (d > maxd) maxd = d; }
    printf("variant %d: mismatches %d / 1024, max diff %.3f (ref[0]=%.1f out[0]=%.1f)\n", variant, bad, maxd, ref[0], out[0]);
  }
  return 0;
}
