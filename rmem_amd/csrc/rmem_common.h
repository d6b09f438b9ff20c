// rmem_common.h -- shared device helpers for the RMem gfx950 kernels.
//
// Numeric convention ("split-fp16"): an fp32 value x is carried as two fp16 planes
// hi = fp16(x), lo = fp16(x - hi): 22 significant bits (fp16 subnormals are kept by the
// conversions and by the MFMA, research/ubench/f16_denorm.hip).  A product of two such values
// evaluated on the 16-bit MFMA pipe as  hi*lo' + lo*hi' + hi*hi'  (NSPLIT = 3) has a relative
// error of ~2^-21 per product, i.e. fp32-class, at 3 MFMA issues; NSPLIT = 1 uses hi*hi' only
// (plain fp16).  All accumulation is fp32 in the MFMA accumulators.  (bf16 hi/lo planes carry 16
// bits, 2^-17 per product: measured on the golden 480p clip that rounding alone moved 31 label
// pixels over 9 frames, fp16 planes move 3 -- tools/precision_study.py lin16bf / lin16x2 -- at the
// same MFMA rate.)  Values beyond +-65504 saturate (the reference itself evaluates under fp16
// autocast with --amp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RMEM_OK 0
#define RMEM_ERR_INVALID (-1)
#define RMEM_ERR_LAUNCH (-2)

// Process-wide tuning / debug switches of the library.  The library never reads the environment: a host sets these through
// rmem_configure() (include/rmem_hip.h; rmem_amd/hip.py maps its RMEM_* variables onto it once, when it loads the library).
struct RmemConfig {
  int linear_tiles = 0;      // 1: tile-per-workgroup projection kernels for every launch (A/B and cross-check)
  int stream_var = 1;        // rmem_linear_trace only: 2 no operand requests, 3 no MFMAs, 4 no fragment reads (timing experiments)
  int dw_rx = 9, dw_v = 1;   // depth-wise conv, one-row kernel: tokens per thread, channels per thread
  int dw_rows = 2;           // depth-wise conv: output rows per thread (0: the one-row kernel)
  int dw_grid_order = 0;     // 1: plain grid order instead of the XCD-aware block order
  int ida_tokens = 1, ida_unroll = 16;   // ID assignment: tokens per block, unroll
  int read_var = 0;          // rmem_attn_read_trace only: experiment mask of the tracing kernel (4, 8, 16)
};
RmemConfig& rmem_config();   // batch.hip

typedef unsigned short h16_t;  // raw 16 bits of a plane element (IEEE fp16)
typedef __attribute__((ext_vector_type(8))) _Float16 frag8_t;   // MFMA operand fragment: 8 consecutive k of one row
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// native vector (HIP's uint4 struct keeps register arrays in scratch)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#include <stdio.h>
#define RMEM_CHECK_LAUNCH()                                                                          \
  do {                                                                                               \
    hipError_t e__ = hipGetLastError();                                                              \
    if (e__ != hipSuccess) {                                                                         \
      fprintf(stderr, "rmem_hip: launch failed at %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return RMEM_ERR_LAUNCH;                                                                        \
    }                                                                                                \
  } while (0)

// fp16 bits of x, round-to-nearest-even (subnormals kept)
__device__ __forceinline__ unsigned short f2h_bits(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float h_bits2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

__device__ __forceinline__ void split_f16(float x, h16_t& hi, h16_t& lo) {   // -> fp16 hi/lo planes
  const float xc = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  hi = f2h_bits(xc);
  lo = f2h_bits(__builtin_amdgcn_fmed3f(xc - h_bits2f(hi), -65504.f, 65504.f));
}
// Two fp32 values -> their packed fp16 hi / lo planes (hi = fp16(x), lo = fp16(x - hi), round to nearest even) in three
// instructions: one v_cvt_pk_f16_f32 and two mixed-precision fmas that take hi straight from its fp16 half and write
// the rounded remainder into the low / high half of `lo` (x - float(hi) is exact in fp32, so the bits are those of the
// convert / subtract / convert form the compiler emits -- which is five instructions per pair; it does not form
// v_fma_mix itself).
__device__ __forceinline__ void split_pair_f16(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t_;
  f32x2_t_ pp;
  pp[0] = x0;
  pp[1] = x1;
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(pp, f16x2_t));
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(x0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(hi));
}

// x of lane L and of lane L ^ 32 in ONE instruction (v_permlane32_swap; __shfl_xor(x, 32) is an LDS round trip plus the
// index arithmetic): lo = x of the lane's copy in lanes 0-31, hi = x of its copy in lanes 32-63, the same pair in both
// lanes -- so max(lo, hi) / lo + hi are the xor-32 reductions, identical in the two lanes.  (The elements of the
// builtin's result are copied to scalars before the bit cast: a bit cast applied to `r[1]` directly reads element 0.)
__device__ __forceinline__ void xor32_pair(float x, float& lo, float& hi) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const uint32_t r0 = r[0], r1 = r[1];
  lo = __builtin_bit_cast(float, r0);
  hi = __builtin_bit_cast(float, r1);
}

// the same for lanes L and L ^ 16 (v_permlane16_swap: odd 16-lane rows of one operand against even rows of the other):
// lo = x of the pair's copy in the even row, hi = in the odd row, the same pair in both lanes
__device__ __forceinline__ void xor16_pair(float x, float& lo, float& hi) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const uint32_t r0 = r[0], r1 = r[1];
  lo = __builtin_bit_cast(float, r0);
  hi = __builtin_bit_cast(float, r1);
}

// x of another lane of the 16-lane row through the DPP crossbar (a VALU move the compiler folds into the consuming add;
// no LDS round trip): CTRL = 0x128 row_ror:8 (lane ^ 8), 0x141 row_half_mirror (7 - lane % 8), quad_perm 0x1B [3,2,1,0]
// (lane ^ 3), 0x4E [2,3,0,1] (lane ^ 2), 0xB1 [1,0,3,2] (lane ^ 1)
template <int CTRL>
__device__ __forceinline__ float dpp_lane(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

// The value of lane ^ O for O = 32, 16 (lane-swap + select), 8, 4, 2, 1 (DPP): drop-in for __shfl_xor(x, O) without the
// ds_bpermute round trip; doubles go as two halves.
template <int O>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t u) {
  static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "power of two below 64");
  if constexpr (O == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    return (__lane_id() & 32) ? r0 : r1;
  } else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    return (__lane_id() & 16) ? r0 : r1;
  } else {
    const int v = (int)u;
    if constexpr (O == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    else if constexpr (O == 4)
      return (uint32_t)__builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false), 0x1B, 0xf, 0xf, false);
    else if constexpr (O == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
  }
}
template <int O>
__device__ __forceinline__ float xor_lane(float x) {
  return __builtin_bit_cast(float, xor_lane_u32<O>(__builtin_bit_cast(uint32_t, x)));
}
template <int O>
__device__ __forceinline__ double xor_lane(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned long long r = ((unsigned long long)xor_lane_u32<O>((uint32_t)(u >> 32)) << 32) | xor_lane_u32<O>((uint32_t)u);
  return __builtin_bit_cast(double, r);
}
// s += s[lane ^ o] for o = FROM, FROM / 2, ..., 1 (the __shfl_xor butterfly, same partners in the same order)
template <int FROM = 32, class T>
__device__ __forceinline__ T wave_sum_xor_t(T s) {
  if constexpr (FROM >= 32) s += xor_lane<32>(s);
  if constexpr (FROM >= 16) s += xor_lane<16>(s);
  if constexpr (FROM >= 8) s += xor_lane<8>(s);
  if constexpr (FROM >= 4) s += xor_lane<4>(s);
  if constexpr (FROM >= 2) s += xor_lane<2>(s);
  s += xor_lane<1>(s);
  return s;
}
__device__ __forceinline__ float wave_max_xor(float m) {
  m = fmaxf(m, xor_lane<32>(m));
  m = fmaxf(m, xor_lane<16>(m));
  m = fmaxf(m, xor_lane<8>(m));
  m = fmaxf(m, xor_lane<4>(m));
  m = fmaxf(m, xor_lane<2>(m));
  return fmaxf(m, xor_lane<1>(m));
}

// Sum over the 64 lanes, every lane gets it: the butterfly  s += s[lane ^ o]  for o = 32, 16, 8, 4, 2, 1 -- bit for bit
// the chain of __shfl_xor steps it replaces (fp32 addition is commutative; the partner at every step is the same lane) --
// on lane-swap and DPP instructions instead of twelve dependent ds_bpermute round trips per row (~1.5 k cycles per
// LayerNorm row, the whole cost of the small normalisation kernels: profiles/r05b_kbench_rowres.json).
__device__ __forceinline__ float wave_sum_xor(float s) {
  float lo, hi;
  xor32_pair(s, lo, hi);
  s = lo + hi;
  xor16_pair(s, lo, hi);
  s = lo + hi;
  s += dpp_lane<0x128>(s);
  s += dpp_lane<0x1B>(dpp_lane<0x141>(s));    // lane ^ 4 = (lane ^ 7) ^ 3
  s += dpp_lane<0x4E>(s);
  s += dpp_lane<0xB1>(s);
  return s;
}

#define RMEM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// Monotone float <-> uint encoding so that atomicMax on the uint orders like the
// float; 0 encodes "below every float" (lets a plain memset(0) reset the buffer).
__device__ __forceinline__ uint32_t enc_ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// exp for softmax weights exp(s - max), s - max <= 0: v_exp_f32(x * log2(e)), 2 instructions
// instead of the 13 of the correctly rounded expf (the flash MHA kernel is VALU-bound: 91 vs
// 125 us).  Relative error ~ |x| * 6e-8 + 1 ulp: below 1e-6 for every weight that matters
// (x > -15), an order of magnitude inside the split-fp16 product error (2^-21 relative per product); carrying the
// rounding error of the product as a correction term (6 instructions) changed nothing measurable.
// -3e38 sentinels overflow to -inf and give exactly 0.
__device__ __forceinline__ float exp_weight(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// SiLU x / (1 + e^-x) in ten instructions (the correctly rounded expf + IEEE division form is ~50, sixteen times per
// lane in the epilogue of every V / U projection: 3 k of the 6 k cycles of such an epilogue, profiles/r04f_stream_trace.json).
// e^-x = 2^(t + tl): t = fl(-x log2 e), tl = the rounding error of that product + the low part of log2 e, applied as the
// factor (1 + tl ln 2) behind v_exp_f32 (1 ulp); 1 / d by v_rcp_f32 + one Newton step.  ~3 ulp (2e-7 relative) against
// ~1.5 ulp before -- inside the 2^-21 of one split-fp16 product.  t is clamped so that d stays finite (x < -83: the true
// value is below 1e-34 in magnitude either way).
__device__ __forceinline__ float silu_f(float x) {
  const float nx = -x;
  const float tu = nx * 1.44269504088896341f;
  const float tl = fmaf(nx, 1.92596299112661746e-8f, fmaf(nx, 1.44269504088896341f, -tu));
  const float t = fminf(tu, 120.0f);
  float e = __builtin_amdgcn_exp2f(t);
  e = fmaf(e, tl * 0.693147180559945f, e);
  const float d = 1.0f + e;
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  return x * r;
}

// LayerNorm of one row of 256 held as a float4 per lane (64 lanes): the normalised values of the lane's four columns.  ONE
// definition for every kernel that normalises a row (pointwise.hip: ln_row256; linear_rowres.h), so that they agree bit for
// bit: every multiply-add is spelled as fmaf (left to the compiler the contraction depended on whether the SLP vectoriser
// got to the products first), the mean is a product with a power of two (exact, contraction cannot change it).
__device__ __forceinline__ void ln_row256_gb(const float4 v, const float4 g, const float4 b, float eps, float (&y)[4]) {
  const float s = wave_sum_xor(v.x + v.y + v.z + v.w);
  const float mean = s * (1.0f / 256.0f);
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  const float ss = wave_sum_xor(fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0))));
  const float rstd = 1.0f / sqrtf(fmaf(ss, 1.0f / 256.0f, eps));
  y[0] = fmaf(d0 * rstd, g.x, b.x);
  y[1] = fmaf(d1 * rstd, g.y, b.y);
  y[2] = fmaf(d2 * rstd, g.z, b.z);
  y[3] = fmaf(d3 * rstd, g.w, b.w);
}
__device__ __forceinline__ void ln_row256_vals(const float4 v, const float* gamma, const float* beta, float eps, int lane,
                                               float (&y)[4]) {
  ln_row256_gb(v, *reinterpret_cast<const float4*>(gamma + lane * 4), *reinterpret_cast<const float4*>(beta + lane * 4), eps, y);
}

// exact floor(k / w) for 0 <= k < 2^20, 1 <= w ((k+0.5)/w is never
// closer than 0.5/w to an integer, far above fp32 rounding).
__device__ __forceinline__ int fast_div(int k, float inv_w) {
  return (int)(((float)k + 0.5f) * inv_w);
}
