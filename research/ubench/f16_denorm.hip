// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal operands, and does the fp32->fp16 conversion produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ void k(float* out, float tiny) {
  f16x8_t a, b;
  const _Float16 t = (_Float16)tiny;            // conversion
  for (int i = 0; i < 8; ++i) { a[i] = (i == 0 && (threadIdx.x >> 5) == 0) ? t : (_Float16)0.f; b[i] = (_Float16)1.0f; }
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)t; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  for (float tiny : {1e-3f, 3e-5f, 1e-6f, 1e-7f, 6e-8f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("x = %.3e: fp16(x) = %.6e, mfma(x * 1) = %.6e\n", tiny, h[1], h[0]);
  }
  return 0;
}
