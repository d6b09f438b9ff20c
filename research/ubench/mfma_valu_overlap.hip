// Micro-benchmark: do MFMA work and VALU work of DIFFERENT waves on one SIMD overlap?
// Every wave loops over "tiles": a chain of NM dependent v_mfma_f32_32x32x16_f16 (one accumulator, VGPR form), then NV
// dependent-ish VALU instructions (4 fma chains) that consume the accumulator and produce the next B operand --
// the shape of a flash-attention tile (scores -> softmax -> P.V).  Modes: MFMA only, VALU only, both.  With W waves per
// SIMD (W blocks of 4 waves per CU) perfect overlap gives time(both) = max(time(mfma), time(valu)); no overlap gives the sum.
// A second variant issues the MFMAs on TWO alternating accumulators (independent back-to-back MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) f16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NM, int NV, int NACC, int PRIO, int OFFSET>
__global__ __launch_bounds__(256, 3) void tile_loop(float* out, int iters, float seed) {
  f16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * ((threadIdx.x & 63) + i)); b[i] = (f16)(0.002f * (i + 1)); }
  f32x16_t acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float v0 = seed, v1 = seed + 1.f, v2 = seed + 2.f, v3 = seed + 3.f;
  if (OFFSET && NV > 0 && ((blockIdx.x / 256) & 1)) {          // every other resident block starts half a tile later (one VALU phase)
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) {
      v0 = fmaf(v0, 1.0001f, v1); v1 = fmaf(v1, 0.9999f, v2); v2 = fmaf(v2, 1.0002f, v3); v3 = fmaf(v3, 0.9998f, v0);
    }
  }
  for (int it = 0; it < iters; ++it) {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (NV > 0) {
      if (NM > 0) { v0 += acc[0][0]; v1 += acc[0][5]; v2 += acc[NACC - 1][10]; v3 += acc[NACC - 1][15]; }
#pragma unroll
      for (int k = 0; k < NV / 4; ++k) {
        v0 = fmaf(v0, 1.0001f, v1); v1 = fmaf(v1, 0.9999f, v2); v2 = fmaf(v2, 1.0002f, v3); v3 = fmaf(v3, 0.9998f, v0);
      }
      if (NM > 0) b[0] = (f16)(v0 + v1 + v2 + v3);          // the next tile's B operand depends on the VALU phase
    }
  }
  float s = v0 + v1 + v2 + v3;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 123.456f) out[0] = s;
}

template <int NM, int NV, int NACC, int PRIO = 0, int OFFSET = 0>
double run(int waves_per_simd, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;
  tile_loop<NM, NV, NACC, PRIO, OFFSET><<<blocks, 256>>>(out, iters, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    tile_loop<NM, NV, NACC, PRIO, OFFSET><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  hipFree(out);
  return best * 1e3 / iters * 1e3 / waves_per_simd;     // ns per tile per SIMD-slot (= per wave-tile / waves sharing the SIMD)
}

int main() {
  const int iters = 4000;
  printf("ns per tile per SIMD (lower = better); tile = 12 MFMAs 32x32x16 + 104 VALU\n");
  for (int w = 1; w <= 3; ++w) {
    const double m1 = run<12, 0, 1>(w, iters), m2 = run<12, 0, 2>(w, iters), v = run<0, 104, 1>(w, iters);
    const double b1 = run<12, 104, 1>(w, iters), b2 = run<12, 104, 2>(w, iters);
    const double bp = run<12, 104, 1, 1, 0>(w, iters), bo = run<12, 104, 1, 0, 1>(w, iters), bpo = run<12, 104, 1, 1, 1>(w, iters);
    printf("%d wave(s)/SIMD: MFMA-only dependent %.1f | MFMA-only 2 accumulators %.1f | VALU-only %.1f | both (dependent) %.1f | both (2 acc) %.1f"
           " | both + s_setprio around the MFMAs %.1f | both + odd blocks half a tile out of phase %.1f | both + prio + phase %.1f   [sum %.1f, max %.1f]\n",
           w, m1, m2, v, b1, b2, bp, bo, bpo, m1 + v, m1 > v ? m1 : v);
  }
  return 0;
}
