// Micro-benchmark (round 6): how much INDEPENDENT VALU work hides beside v_mfma_f32_32x32x16_f16 on a gfx950 SIMD?
//
// research/ubench/mfma_valu_overlap.hip (round 4) alternated two DEPENDENT phases per wave (12 chained MFMAs, then 104
// VALU that consume the accumulator and produce the next B operand) and asked whether OTHER waves fill the idle pipe:
// they did not (time = sum).  It never placed independent VALU work between the MFMAs of ONE wave, which is what the
// hardware guide measured (MI355X_MICROARCH.md: <= 5 single-issue instructions hidden per 32x32x16 gap).  This file does:
//
//   A. intra-wave: a tile = 12 x (one MFMA on the same accumulator + K independent fillers), the stream fixed by inline
//      assembly (the compiler cannot re-order it).  Fillers: v_fma_f32 on four independent chains, or v_exp_f32, or
//      v_cvt_pk_f16_f32 -- the softmax's instructions.  K = 0 .. 8, one and two waves per SIMD.
//   B. cross-wave, NO dependence between the streams: waves 0-3 of a 512-thread block run MFMAs only, waves 4-7 (their SIMD
//      partners) run VALU only, each a fixed amount of work; the block's time against each half alone.
//
// Output: ns per tile per SIMD and the cycles per MFMA they imply at the measured clock (MFMA-only = 32 cycles each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) f16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define MF "v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n\t"
#define FA "v_fma_f32 %1, %1, %7, %8\n\t"
#define FB "v_fma_f32 %2, %2, %7, %8\n\t"
#define FC "v_fma_f32 %3, %3, %7, %8\n\t"
#define FD "v_fma_f32 %4, %4, %7, %8\n\t"
#define EA "v_exp_f32 %1, %1\n\t"
#define EB "v_exp_f32 %2, %2\n\t"
#define EC "v_exp_f32 %3, %3\n\t"
#define ED "v_exp_f32 %4, %4\n\t"
#define CA "v_cvt_pk_f16_f32 %1, %2, %3\n\t"
#define CB "v_cvt_pk_f16_f32 %4, %2, %3\n\t"

// K fillers of kind T behind each MFMA (T: 0 fma, 1 exp, 2 cvt_pk)
template <int K, int T>
__device__ __forceinline__ void gap(f32x16_t& acc, float& v0, float& v1, float& v2, float& v3, const f16x8_t& a, const f16x8_t& b,
                                    float c0, float c1) {
#define OPS : "+v"(acc), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b), "v"(c0), "v"(c1)
  if constexpr (T == 0) {
    if constexpr (K == 0) asm volatile(MF OPS);
    if constexpr (K == 1) asm volatile(MF FA OPS);
    if constexpr (K == 2) asm volatile(MF FA FB OPS);
    if constexpr (K == 3) asm volatile(MF FA FB FC OPS);
    if constexpr (K == 4) asm volatile(MF FA FB FC FD OPS);
    if constexpr (K == 5) asm volatile(MF FA FB FC FD FA OPS);
    if constexpr (K == 6) asm volatile(MF FA FB FC FD FA FB OPS);
    if constexpr (K == 7) asm volatile(MF FA FB FC FD FA FB FC OPS);
    if constexpr (K == 8) asm volatile(MF FA FB FC FD FA FB FC FD OPS);
  } else if constexpr (T == 1) {
    if constexpr (K == 1) asm volatile(MF EA OPS);
    if constexpr (K == 2) asm volatile(MF EA EB OPS);
    if constexpr (K == 3) asm volatile(MF EA EB EC OPS);
    if constexpr (K == 4) asm volatile(MF EA EB EC ED OPS);
    if constexpr (K == 5) asm volatile(MF EA EB EC ED EA OPS);
    if constexpr (K == 6) asm volatile(MF EA EB EC ED EA EB OPS);
  } else {
    if constexpr (K == 2) asm volatile(MF CA CB OPS);
    if constexpr (K == 4) asm volatile(MF CA CB CA CB OPS);
    if constexpr (K == 6) asm volatile(MF CA CB CA CB CA CB OPS);
  }
#undef OPS
}

template <int K, int T>
__global__ __launch_bounds__(256, 2) void interleaved(float* out, int iters, float seed) {
  f16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * ((threadIdx.x & 63) + i)); b[i] = (f16)(0.002f * (i + 1)); }
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v0 = seed, v1 = seed * 0.5f, v2 = seed * 0.25f, v3 = seed * 0.125f;
  const float c0 = 0.5f, c1 = -0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) gap<K, T>(acc, v0, v1, v2, v3, a, b, c0, c1);
  }
  float s = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[0] = s;
}

// the same fillers with no MFMA (what the VALU work costs alone)
template <int K>
__global__ __launch_bounds__(256, 2) void valu_only(float* out, int iters, float seed) {
  float v0 = seed, v1 = seed * 0.5f, v2 = seed * 0.25f, v3 = seed * 0.125f;
  const float c0 = 0.5f, c1 = -0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12 * K / 4; ++m)
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c0), "v"(c1));
  }
  const float s = v0 + v1 + v2 + v3;
  if (s == 123.456f) out[0] = s;
}

// B: waves 0-3 MFMA only (12 per tile), waves 4-7 VALU only (NV fmas per tile); wave w and w + 4 share a SIMD.
// WHICH: 1 = only the MFMA half works (the other exits), 2 = only the VALU half, 3 = both.
template <int NV, int WHICH>
__global__ __launch_bounds__(512, 1) void split_roles(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * ((threadIdx.x & 63) + i)); b[i] = (f16)(0.002f * (i + 1)); }
  float s = 0.f;
  if (wave < 4) {
    if (!(WHICH & 1)) return;
    f32x16_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 12; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
    for (int r = 0; r < 16; ++r) s += acc[r];
  } else {
    if (!(WHICH & 2)) return;
    float v0 = seed, v1 = seed * 0.5f, v2 = seed * 0.25f, v3 = seed * 0.125f;
    const float c0 = 0.5f, c1 = -0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < NV / 4; ++m)
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c0), "v"(c1));
    }
    s = v0 + v1 + v2 + v3;
  }
  if (s == 123.456f) out[0] = s;
}

template <class F>
double time_ns_per_tile(F launch, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best * 1e6 / iters;
}

template <int K, int T>
void row(float* out, int iters, double base1, double base2) {
  const double t1 = time_ns_per_tile([&] { interleaved<K, T><<<256, 256>>>(out, iters, 0.5f); }, iters);
  const double t2 = time_ns_per_tile([&] { interleaved<K, T><<<512, 256>>>(out, iters, 0.5f); }, iters) / 2;
  const char* kind = T == 0 ? "v_fma_f32" : (T == 1 ? "v_exp_f32" : "v_cvt_pk_f16_f32");
  printf("| %d x %s | %.1f | %.2f | %.1f | %.2f |\n", K, kind, t1, 32.0 * t1 / base1, t2, 32.0 * t2 / base2);
}

int main() {
  const int iters = 4000;
  float* out; hipMalloc(&out, 4);
  printf("A. one wave's stream: 12 x (MFMA 32x32x16 f16 on one accumulator + K independent fillers); ns per tile per SIMD\n");
  const double b1 = time_ns_per_tile([&] { interleaved<0, 0><<<256, 256>>>(out, iters, 0.5f); }, iters);
  const double b2 = time_ns_per_tile([&] { interleaved<0, 0><<<512, 256>>>(out, iters, 0.5f); }, iters) / 2;
  printf("MFMA only: %.1f ns per tile with one wave per SIMD, %.1f with two (= 12 x 32 cycles: clock %.2f GHz)\n", b1, b2, 384.0 / b1);
  printf("| fillers per MFMA | 1 wave/SIMD ns | cycles per MFMA | 2 waves/SIMD ns | cycles per MFMA |\n|---|---|---|---|---|\n");
  row<1, 0>(out, iters, b1, b2); row<2, 0>(out, iters, b1, b2); row<3, 0>(out, iters, b1, b2); row<4, 0>(out, iters, b1, b2);
  row<5, 0>(out, iters, b1, b2); row<6, 0>(out, iters, b1, b2); row<7, 0>(out, iters, b1, b2); row<8, 0>(out, iters, b1, b2);
  row<2, 1>(out, iters, b1, b2); row<4, 1>(out, iters, b1, b2); row<6, 1>(out, iters, b1, b2);
  row<2, 2>(out, iters, b1, b2); row<4, 2>(out, iters, b1, b2); row<6, 2>(out, iters, b1, b2);
  const double v4 = time_ns_per_tile([&] { valu_only<4><<<256, 256>>>(out, iters, 0.5f); }, iters);
  const double v8 = time_ns_per_tile([&] { valu_only<8><<<256, 256>>>(out, iters, 0.5f); }, iters);
  printf("the fillers alone (no MFMA): 48 fmas %.1f ns, 96 fmas %.1f ns per tile\n\n", v4, v8);

  printf("B. independent streams on SIMD partners (waves w: 12 MFMAs per tile, waves w + 4: NV fmas per tile), ns per tile\n");
  printf("| NV | MFMA half alone | VALU half alone | both | sum | max |\n|---|---|---|---|---|---|\n");
  {
    const double m = time_ns_per_tile([&] { split_roles<48, 1><<<256, 512>>>(out, iters, 0.5f); }, iters);
    const double v = time_ns_per_tile([&] { split_roles<48, 2><<<256, 512>>>(out, iters, 0.5f); }, iters);
    const double bth = time_ns_per_tile([&] { split_roles<48, 3><<<256, 512>>>(out, iters, 0.5f); }, iters);
    printf("| 48 | %.1f | %.1f | %.1f | %.1f | %.1f |\n", m, v, bth, m + v, m > v ? m : v);
  }
  {
    const double m = time_ns_per_tile([&] { split_roles<96, 1><<<256, 512>>>(out, iters, 0.5f); }, iters);
    const double v = time_ns_per_tile([&] { split_roles<96, 2><<<256, 512>>>(out, iters, 0.5f); }, iters);
    const double bth = time_ns_per_tile([&] { split_roles<96, 3><<<256, 512>>>(out, iters, 0.5f); }, iters);
    printf("| 96 | %.1f | %.1f | %.1f | %.1f | %.1f |\n", m, v, bth, m + v, m > v ? m : v);
  }
  {
    const double m = time_ns_per_tile([&] { split_roles<192, 1><<<256, 512>>>(out, iters, 0.5f); }, iters);
    const double v = time_ns_per_tile([&] { split_roles<192, 2><<<256, 512>>>(out, iters, 0.5f); }, iters);
    const double bth = time_ns_per_tile([&] { split_roles<192, 3><<<256, 512>>>(out, iters, 0.5f); }, iters);
    printf("| 192 | %.1f | %.1f | %.1f | %.1f | %.1f |\n", m, v, bth, m + v, m > v ? m : v);
  }
  hipFree(out);
  return 0;
}
