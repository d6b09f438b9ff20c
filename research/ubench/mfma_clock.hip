// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate with every SIMD busy (no memory traffic).
// Tells the effective MFMA clock under full-chip matrix load, i.e. the real ceiling for pv_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
  f32x16_t acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run(int blocks, int iters, const char* tag) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double mf = (double)blocks * 4 * iters * NACC;           // wave-MFMAs
  double tf = mf * 32768.0 / (best * 1e-3) / 1e12;
  double cyc_per_mfma_at_2p4 = best * 1e-3 * 2.4e9 / ((double)iters * NACC * ((blocks + 255) / 256));
  printf("%s: blocks %d iters %d nacc %d  %.3f ms  %.0f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", tag, blocks,
         iters, NACC, best, tf, cyc_per_mfma_at_2p4);
}

int main() {
  run<4>(256, 20000, "1 wave/SIMD, 4 independent accumulators");
  run<4>(512, 20000, "2 waves/SIMD, 4 independent accumulators");
  run<1>(512, 20000, "2 waves/SIMD, 1 accumulator (dependent chain)");
  run<4>(64, 20000, "quarter of the CUs");
  run<4>(256, 2000, "short burst (70 us)");
  return 0;
}
