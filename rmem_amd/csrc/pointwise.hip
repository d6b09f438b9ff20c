// pointwise.hip -- bandwidth-bound kernels of the hot path: LayerNorm -> planes,
// depth-wise 5x5 -> planes, final GroupNorm, ID assignment (gather-conv + LayerNorm),
// RMem attention-mass reduction, fp32 -> planes.  See include/rmem_hip.h.
#include "../../include/rmem_hip.h"
#include "rmem_common.h"
#include "launch.h"
#include <stdlib.h>

// ------------------------------------------------------------------ LayerNorm -> planes
// one wave per row, C = 256: 4 consecutive channels per lane (16-byte loads).
// y = LN(x + x2) * gamma + beta + post  (x2, post optional)
__device__ __forceinline__ void layernorm_split_row(const float* x, long ldx, const float* x2, long ldx2,
                                                    const float* gamma, const float* beta, int N,
                                                    float eps, const float* post, long ldpost,
                                                    h16_t* oh, h16_t* ol, long ldo, float* of32,
                                                    long ldof, float* sum_out = nullptr, long ldsum = 0) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  float4 v = *reinterpret_cast<const float4*>(x + (long)row * ldx + lane * 4);
  if (x2) {
    const float4 w = *reinterpret_cast<const float4*>(x2 + (long)row * ldx2 + lane * 4);
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    if (sum_out) *reinterpret_cast<float4*>(sum_out + (long)row * ldsum + lane * 4) = v;   // (a row is read and written by one wave)
  }
  float s = v.x + v.y + v.z + v.w;
  s = wave_sum_xor_t(s);
  const float mean = s * (1.0f / 256.0f);
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  float ss = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  ss = wave_sum_xor_t(ss);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + eps);
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float4 b = *reinterpret_cast<const float4*>(beta + lane * 4);
  float y[4] = {d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w};
  if (post) {
    const float4 w = *reinterpret_cast<const float4*>(post + (long)row * ldpost + lane * 4);
    y[0] += w.x; y[1] += w.y; y[2] += w.z; y[3] += w.w;
  }
  if (of32) *reinterpret_cast<float4*>(of32 + (long)row * ldof + lane * 4) = make_float4(y[0], y[1], y[2], y[3]);
  if (oh) {
    h16_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_f16(y[e], hi[e], lo[e]);
    uint2 vh, vl;
    vh.x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
    vh.y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
    vl.x = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
    vl.y = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
    *reinterpret_cast<uint2*>(oh + (long)row * ldo + lane * 4) = vh;
    if (ol) *reinterpret_cast<uint2*>(ol + (long)row * ldo + lane * 4) = vl;
  }
}

__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* x, long ldx, const float* x2, long ldx2,
                                                              const float* gamma, const float* beta, int N,
                                                              float eps, const float* post, long ldpost,
                                                              h16_t* oh, h16_t* ol, long ldo, float* of32,
                                                              long ldof) {
  layernorm_split_row(x, ldx, x2, ldx2, gamma, beta, N, eps, post, ldpost, oh, ol, ldo, of32, ldof);
}

// up to four such LayerNorms over the same N rows in ONE launch (blockIdx.y = problem): the AOT block normalises one
// input twice (with and without the positional embedding) and two inputs with one norm -- each pair was two launches
struct LnMultiArgs {
  rmem_ln_args p[4];
  int N;
  float eps;
};
__global__ __launch_bounds__(256) void layernorm_multi_kernel(LnMultiArgs g) {
  const rmem_ln_args& a = g.p[blockIdx.y];
  layernorm_split_row(a.x, (long)a.ldx, a.x2, (long)a.ldx2, a.gamma, a.beta, g.N, g.eps, a.post, (long)a.ldpost,
                      reinterpret_cast<h16_t*>(a.oh), reinterpret_cast<h16_t*>(a.ol), (long)a.ldo, a.of32, (long)a.ldof,
                      a.sum_out, (long)a.ldsum);
}

extern "C" int rmem_layernorm_ex(const float* x, int64_t ldx, const float* x2, int64_t ldx2, const float* gamma,
                                 const float* beta, int32_t N, int32_t C, float eps, const float* post,
                                 int64_t ldpost, rmem_f16* oh, rmem_f16* ol, int64_t ldo, float* of32,
                                 int64_t ldof, void* stream) {
  if (!x || !gamma || !beta || N <= 0 || C != 256 || (ldx % 4) || (ldo % 4) || (ldof % 4) || (ldx2 % 4) ||
      (ldpost % 4))
    return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(layernorm_split_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     (long)ldx, x2, (long)ldx2, gamma, beta, N, eps, post, (long)ldpost, oh, ol, (long)ldo, of32,
                     (long)ldof);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_layernorm_multi(const rmem_ln_args* p, int32_t n, int32_t N, int32_t C, float eps, void* stream) {
  if (!p || n <= 0 || n > 4 || N <= 0 || C != 256) return RMEM_ERR_INVALID;
  LnMultiArgs g;
  for (int i = 0; i < n; ++i) {
    const rmem_ln_args& a = p[i];
    if (!a.x || !a.gamma || !a.beta || (a.ldx % 4) || (a.ldo % 4) || (a.ldof % 4) || (a.ldx2 % 4) || (a.ldpost % 4) ||
        (a.ldsum % 4) || (a.sum_out && !a.x2))
      return RMEM_ERR_INVALID;
    g.p[i] = a;
  }
  g.N = N;
  g.eps = eps;
  hipLaunchKernelGGL(layernorm_multi_kernel, dim3((N + 3) / 4, n), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// LayerNorm of one row of 256 held as a float4 per lane -> planes (and optionally fp32): ONE definition for every kernel
// that normalises a row, so that they agree bit for bit (same expression tree, same contraction).
__device__ __forceinline__ void ln_row256(const float4 v, const float* gamma, const float* beta, float eps, int lane, long row,
                                          h16_t* oh, h16_t* ol, long ldo, float* of32, long ldof) {
  float y[4];
  ln_row256_vals(v, gamma, beta, eps, lane, y);     // (rmem_common.h: the one row function, shared with linear_rowres.h)
  if (of32) *reinterpret_cast<float4*>(of32 + row * ldof + lane * 4) = make_float4(y[0], y[1], y[2], y[3]);
  if (oh) {
    h16_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_f16(y[e], hi[e], lo[e]);
    uint2 vh, vl;
    vh.x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
    vh.y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
    vl.x = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
    vl.y = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
    *reinterpret_cast<uint2*>(oh + row * ldo + lane * 4) = vh;
    if (ol) *reinterpret_cast<uint2*>(ol + row * ldo + lane * 4) = vl;
  }
}

// residual reduce (split-K partials, fixed order) + LayerNorm
__device__ __forceinline__ void layernorm_red_body(float* x, long ldx, const float* parts, int nparts,
                                                   long part_stride, long ldpart, const float* gamma,
                                                   const float* beta, int N, float eps, h16_t* oh,
                                                   h16_t* ol, long ldo, float* of32, long ldof) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  float4 v = *reinterpret_cast<const float4*>(x + (long)row * ldx + lane * 4);
  for (int z = 0; z < nparts; ++z) {
    const float4 w = *reinterpret_cast<const float4*>(parts + (long)z * part_stride + (long)row * ldpart + lane * 4);
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  if (nparts > 0) *reinterpret_cast<float4*>(x + (long)row * ldx + lane * 4) = v;
  ln_row256(v, gamma, beta, eps, lane, row, oh, ol, ldo, of32, ldof);
}

struct LnRedArgs {
  float* x; long ldx; const float* parts; int nparts; long part_stride, ldpart; const float* gamma; const float* beta;
  int N; float eps; h16_t* oh; h16_t* ol; long ldo; float* of32; long ldof;
};
__device__ void layernorm_red_kernel(const LnRedArgs& a, int) {
  layernorm_red_body(a.x, a.ldx, a.parts, a.nparts, a.part_stride, a.ldpart, a.gamma, a.beta, a.N, a.eps, a.oh, a.ol,
                     a.ldo, a.of32, a.ldof);
}

extern "C" int rmem_layernorm_red(float* x, int64_t ldx, const float* parts, int32_t nparts, int64_t part_stride,
                                  int64_t ldpart, const float* gamma, const float* beta, int32_t N, int32_t C,
                                  float eps, rmem_f16* oh, rmem_f16* ol, int64_t ldo, float* of32, int64_t ldof,
                                  void* stream) {
  if (!x || !gamma || !beta || N <= 0 || C != 256 || (ldx % 4) || (ldo % 4) || (ldof % 4) || nparts < 0 ||
      (nparts > 0 && (!parts || (ldpart % 4) || (part_stride % 4))))
    return RMEM_ERR_INVALID;
  LnRedArgs a{x, (long)ldx, parts, nparts, (long)part_stride, (long)ldpart, gamma, beta, N, eps, oh, ol, (long)ldo, of32,
              (long)ldof};
  return rmem::launch<LnRedArgs, layernorm_red_kernel, 256>(a, dim3((N + 3) / 4), dim3(256), 0,
                                                             static_cast<hipStream_t>(stream));
}

// LayerNorm of the FIRST layer straight from the encoder's feature map: src is channel-major [256][N] (NCHW, batch 1);
// one launch transposes 16 tokens through LDS, writes the token-major residual stream x [N][256], zeroes the second
// stream (tgt_id starts every frame at 0) and emits the normalised planes -- instead of a transposing copy kernel, a
// fill kernel and the LayerNorm launch (the arithmetic per row is layernorm_red_body's: same values, same order).
struct LnCnArgs {
  const float* src; long lds_; float* x; float* zero; const float* gamma; const float* beta; int N; float eps;
  h16_t* oh; h16_t* ol; long ldo;
};
__device__ void layernorm_cn_kernel(const LnCnArgs& a, int) {
  __shared__ float tile[16][256 + 4];
  const int t0 = blockIdx.x * 16, tid = threadIdx.x;
  {                                            // thread = channel: 16 consecutive tokens of its row (64 B), clamped
    const float* row = a.src + (long)tid * a.lds_;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + i < a.N ? t0 + i : a.N - 1;
      tile[i][tid] = row[t];
    }
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {                // (rolled: the rows' arithmetic must not be merged into vector operations)
    const int tl = w * 4 + j, row = t0 + tl;
    if (row >= a.N) continue;
    const float4 v = *reinterpret_cast<const float4*>(&tile[tl][lane * 4]);
    *reinterpret_cast<float4*>(a.x + (long)row * 256 + lane * 4) = v;
    if (a.zero) *reinterpret_cast<float4*>(a.zero + (long)row * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    ln_row256(v, a.gamma, a.beta, a.eps, lane, row, a.oh, a.ol, a.ldo, nullptr, 0);
  }
}

extern "C" int rmem_layernorm_cn(const float* src_cn, int64_t ld_src, float* x, float* zero, const float* gamma,
                                 const float* beta, int32_t N, int32_t C, float eps, rmem_f16* oh, rmem_f16* ol,
                                 int64_t ldo, void* stream) {
  if (!src_cn || !x || !gamma || !beta || !oh || N <= 0 || C != 256 || ld_src < N || (ldo % 4)) return RMEM_ERR_INVALID;
  LnCnArgs a{src_cn, (long)ld_src, x, zero, gamma, beta, N, eps, oh, ol, (long)ldo};
  return rmem::launch<LnCnArgs, layernorm_cn_kernel, 256>(a, dim3((N + 15) / 16), dim3(256), 0,
                                                           static_cast<hipStream_t>(stream));
}

// two independent rows-of-256 problems of the same shape in one launch (blockIdx.y selects):
// norm1 / id_norm1 and norm2 / id_norm2 of a GPM layer (transformer.py:1104,1120,1223-1224)
struct LnRedOne {
  float* x;
  const float* parts;
  const float* gamma;
  const float* beta;
  h16_t* oh;
  h16_t* ol;
  long ldo;
};
struct LnRed2Args {
  LnRedOne p[2];
  long ldx; int nparts; long part_stride, ldpart; int N; float eps;
};
__device__ void layernorm_red2_kernel(const LnRed2Args& a, int) {
  const LnRedOne p = blockIdx.y ? a.p[1] : a.p[0];   // (a select, not an index: the block may be a local copy in SGPRs)
  const long ldx = a.ldx, part_stride = a.part_stride, ldpart = a.ldpart;
  const int nparts = a.nparts, N = a.N;
  const float eps = a.eps;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  float4 v = *reinterpret_cast<const float4*>(p.x + (long)row * ldx + lane * 4);
  // split-K partials, summed in order; four loads in flight per step (the tail repeats the last
  // partial's address and drops the value)
  for (int z0 = 0; z0 < nparts; z0 += 4) {
    float4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int z = z0 + i < nparts ? z0 + i : nparts - 1;
      w[i] = *reinterpret_cast<const float4*>(p.parts + (long)z * part_stride + (long)row * ldpart + lane * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (z0 + i < nparts) { v.x += w[i].x; v.y += w[i].y; v.z += w[i].z; v.w += w[i].w; }
  }
  if (nparts > 0) *reinterpret_cast<float4*>(p.x + (long)row * ldx + lane * 4) = v;
  ln_row256(v, p.gamma, p.beta, eps, lane, row, p.oh, p.ol, p.ldo, nullptr, 0);
}

extern "C" int rmem_layernorm_red2(float* x0, float* x1, int64_t ldx, const float* parts0, const float* parts1,
                                   int32_t nparts, int64_t part_stride, int64_t ldpart, const float* gamma0,
                                   const float* beta0, const float* gamma1, const float* beta1, int32_t N,
                                   int32_t C, float eps, rmem_f16* oh0, rmem_f16* ol0, int64_t ldo0,
                                   rmem_f16* oh1, rmem_f16* ol1, int64_t ldo1, void* stream) {
  if (!x0 || !x1 || !gamma0 || !beta0 || !gamma1 || !beta1 || !oh0 || !oh1 || N <= 0 || C != 256 || (ldx % 4) ||
      (ldo0 % 4) || (ldo1 % 4) || nparts < 0 ||
      (nparts > 0 && (!parts0 || !parts1 || (ldpart % 4) || (part_stride % 4))))
    return RMEM_ERR_INVALID;
  LnRed2Args a{{{x0, parts0, gamma0, beta0, oh0, ol0, (long)ldo0}, {x1, parts1, gamma1, beta1, oh1, ol1, (long)ldo1}},
               (long)ldx, nparts, (long)part_stride, (long)ldpart, N, eps};
  return rmem::launch<LnRed2Args, layernorm_red2_kernel, 256>(a, dim3((N + 3) / 4, 2), dim3(256), 0,
                                                               static_cast<hipStream_t>(stream));
}

extern "C" int rmem_layernorm_split(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                    int32_t N, int32_t C, float eps, rmem_f16* oh, rmem_f16* ol,
                                    int64_t ldo, float* of32, int64_t ldof, void* stream) {
  return rmem_layernorm_ex(x, ldx, nullptr, 0, gamma, beta, N, C, eps, nullptr, 0, oh, ol, ldo, of32, ldof, stream);
}

// ------------------------------------------------------------------ depth-wise 5x5 -> planes
// thread = V channels x a run of RX output tokens along x; the 5 x (RX+4) input window is
// loaded once (sliding-window reuse: 65 loads for 9 outputs instead of 225) and the 25 taps of its
// channels stay in registers.  block = 256 threads = 256 V channels.  V = 1 (90-100 registers, 4-5
// waves per SIMD, every load of a thread in flight at once) beats V = 4 (256 registers, one wave per
// SIMD, one memory round trip per input row): the kernel is latency-bound, not bandwidth-bound.
struct DwOne {
  const float* g;
  const float* wt;
  h16_t* oh;
  h16_t* ol;
};
template <int V> struct DwVec;
template <> struct DwVec<4> { typedef float4 T; };
template <> struct DwVec<2> { typedef float2 T; };
template <> struct DwVec<1> { typedef float T; };

template <int RX, int V>
__device__ __forceinline__ void dwconv5x5_body(const float* g, long ldg, const float* wt, int h, int w, int C,
                                               h16_t* oh, h16_t* ol, long ldo, int bx, int y, int bz) {
  typedef typename DwVec<V>::T VT;
  const int x0 = bx * RX;
  const int c = (bz * 256 + threadIdx.x) * V;
  if (c >= C) return;
  // every load of the thread is issued before the first use (clamped addresses; out-of-image taps are
  // zeroed by a select afterwards): ONE memory round trip per thread instead of one per input row
  float k[25][V];
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    const VT kv = *reinterpret_cast<const VT*>(wt + (long)t * C + c);
    __builtin_memcpy(k[t], &kv, sizeof(VT));
  }
  float v[5][RX + 4][V];
#pragma unroll
  for (int iy = 0; iy < 5; ++iy) {
    const int yy = y + iy - 2;
    const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
    for (int ix = 0; ix < RX + 4; ++ix) {
      const int xx = x0 + ix - 2;
      const int xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
      const VT vv = *reinterpret_cast<const VT*>(g + (long)(yc * w + xc) * ldg + c);
      __builtin_memcpy(v[iy][ix], &vv, sizeof(VT));
    }
  }
  float acc[RX][V];
#pragma unroll
  for (int o = 0; o < RX; ++o)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[o][e] = 0.f;
#pragma unroll
  for (int iy = 0; iy < 5; ++iy) {
    const int yy = y + iy - 2;
    const bool vy = yy >= 0 && yy < h;
#pragma unroll
    for (int ix = 0; ix < RX + 4; ++ix) {
      const int xx = x0 + ix - 2;
      const bool ok = vy && xx >= 0 && xx < w;
#pragma unroll
      for (int o = 0; o < RX; ++o) {
        const int dx = ix - o;  // tap column for output o
        if (dx < 0 || dx > 4) continue;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[o][e] += (ok ? v[iy][ix][e] : 0.f) * k[iy * 5 + dx][e];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < RX; ++o) {
    const int x = x0 + o;
    if (x >= w) continue;
    const long p = (long)y * w + x;
    h16_t hi[V], lo[V];
#pragma unroll
    for (int e = 0; e < V; ++e) split_f16(acc[o][e], hi[e], lo[e]);
    __builtin_memcpy(oh + p * ldo + c, hi, sizeof(hi));
    if (ol) __builtin_memcpy(ol + p * ldo + c, lo, sizeof(lo));
  }
}

// The same arithmetic for RY consecutive output rows per thread: the (RY + 4) x (RX + 4) input window is loaded once, so an
// input element is fetched (RX + 4) / RX * (RY + 4) / RY times (2.9 for 9 x 4) instead of 7.2 -- the one-row kernel's
// re-reads only stay cheap while the neighbouring workgroups' lines survive in the XCD's L2, and in the frame the encoder
// stream's convolutions evict them (27 us per two maps in the frame against 12 isolated, profiles/r05f_bench_x3.json).
// Per output the taps are accumulated in the one-row kernel's order (row by row, left to right): bit-identical.
template <int RX, int RY>
__device__ __forceinline__ void dwconv5x5_rows_body(const float* g, long ldg, const float* wt, int h, int w, int C,
                                                    h16_t* oh, h16_t* ol, long ldo, int bx, int by, int bz) {
  const int x0 = bx * RX, y0 = by * RY;
  const int c = bz * 256 + threadIdx.x;
  if (c >= C) return;
  float k[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) k[t] = wt[(long)t * C + c];
  float v[RY + 4][RX + 4];
#pragma unroll
  for (int iy = 0; iy < RY + 4; ++iy) {
    const int yy = y0 + iy - 2;
    const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
    for (int ix = 0; ix < RX + 4; ++ix) {
      const int xx = x0 + ix - 2;
      const int xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
      v[iy][ix] = g[(long)(yc * w + xc) * ldg + c];
    }
  }
#pragma unroll
  for (int ry = 0; ry < RY; ++ry) {
    const int y = y0 + ry;
    if (y >= h) break;
    float acc[RX];
#pragma unroll
    for (int o = 0; o < RX; ++o) acc[o] = 0.f;
#pragma unroll
    for (int iy = 0; iy < 5; ++iy) {
      const int yy = y + iy - 2;
      const bool vy = yy >= 0 && yy < h;
#pragma unroll
      for (int ix = 0; ix < RX + 4; ++ix) {
        const int xx = x0 + ix - 2;
        const bool ok = vy && xx >= 0 && xx < w;
#pragma unroll
        for (int o = 0; o < RX; ++o) {
          const int dx = ix - o;
          if (dx < 0 || dx > 4) continue;
          acc[o] += (ok ? v[ry + iy][ix] : 0.f) * k[iy * 5 + dx];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < RX; ++o) {
      const int x = x0 + o;
      if (x >= w) continue;
      const long p = (long)y * w + x;
      h16_t hi, lo;
      split_f16(acc[o], hi, lo);
      oh[p * ldo + c] = hi;
      if (ol) ol[p * ldo + c] = lo;
    }
  }
}

// one map (p[0], nmaps = 1) or two maps of the same geometry (the gated long-term and short-term
// aggregates of a layer) in one launch: z = map * nz + channel block
// Block order.  The work items are (z = map x 256-channel block, row y, run of RX tokens); a thread's 5 x (RX + 4) input
// window overlaps its neighbours' (each input is read 7.2 times), and those re-reads only hit in L2 when the neighbours
// run on the SAME XCD -- the eight L2s are separate, and workgroup b goes to XCD b % 8.  In (x, y, z) grid order
// neighbouring runs land on different XCDs and every XCD pulls nearly the whole input from the memory side (8 x 13.7 MB
// for a layer's two maps).  Here XCD k = b % 8 takes the k-th eighth of the items in z-major order: with two maps x four
// channel blocks exactly one z slice each, with one map half the rows of a slice.
struct DwArgs {
  DwOne p[2];
  long ldg; int h, w, C; long ldo; int nz;
  int gx, cap, total, xcd;     // runs per row; items per XCD; items; 0 = plain grid order (RMEM_DW_ORDER=grid, the A/B switch)
};
template <int RX, int V>
__device__ void dwconv5x5_split_kernel(const DwArgs& a, int) {
  const int b = blockIdx.x;
  const int wid = a.xcd ? (b & 7) * a.cap + (b >> 3) : b;
  if (wid >= a.total) return;
  const int per_z = a.gx * a.h;
  const int bz = wid / per_z;
  const int r = wid - bz * per_z;
  const int y = r / a.gx, bx = r - y * a.gx;
  const int which = bz < a.nz ? 0 : 1;
  const DwOne& p = a.p[which];
  dwconv5x5_body<RX, V>(p.g, a.ldg, p.wt, a.h, a.w, a.C, p.oh, p.ol, a.ldo, bx, y, which ? bz - a.nz : bz);
}

template <int RX, int RY>
__device__ void dwconv5x5_rows_kernel(const DwArgs& a, int) {
  const int b = blockIdx.x;
  const int wid = a.xcd ? (b & 7) * a.cap + (b >> 3) : b;
  if (wid >= a.total) return;
  const int gy = (a.h + RY - 1) / RY;
  const int per_z = a.gx * gy;
  const int bz = wid / per_z;
  const int r = wid - bz * per_z;
  const int by = r / a.gx, bx = r - by * a.gx;
  const int which = bz < a.nz ? 0 : 1;
  const DwOne& p = a.p[which];
  dwconv5x5_rows_body<RX, RY>(p.g, a.ldg, p.wt, a.h, a.w, a.C, p.oh, p.ol, a.ldo, bx, by, which ? bz - a.nz : bz);
}

// (RX, V) chosen by measurement; rmem_configure("dw_rx" / "dw_v") overrides (tuning aid); "dw_rows" = RY (2 = default, 3, 4):
// the RY-rows-per-thread kernel with RX = 9; 0 = the one-row kernel
static int launch_dwconv(DwArgs& a, int nmaps, hipStream_t s) {
  const RmemConfig& cfg = rmem_config();
  // default 2: two maps 14.7 -> 11.5 us isolated at 480p, 26.8 -> 19.6 at 720p, 160 -> 146 us per frame in the frame; 3 and 4
  // rows are no faster isolated and slower in the frame (152 registers, one round of 384 workgroups): profiles/r05k_dwconv_rows.txt
  const int ry = cfg.dw_rows;
  if (ry >= 2 && (a.C % 256) == 0) {
    a.nz = a.C / 256;
    a.gx = (a.w + 8) / 9;
    a.total = a.gx * ((a.h + ry - 1) / ry) * nmaps * a.nz;
    a.cap = (a.total + 7) / 8;
    a.xcd = 1;
    const dim3 grid(8 * a.cap);
    if (ry == 2) return rmem::launch<DwArgs, dwconv5x5_rows_kernel<9, 2>, 256>(a, grid, dim3(256), 0, s);
    if (ry == 3) return rmem::launch<DwArgs, dwconv5x5_rows_kernel<9, 3>, 256>(a, grid, dim3(256), 0, s);
    if (ry == 4) return rmem::launch<DwArgs, dwconv5x5_rows_kernel<9, 4>, 256>(a, grid, dim3(256), 0, s);
    return RMEM_ERR_INVALID;
  }
  int rx = cfg.dw_rx, v = cfg.dw_v;   // 480p one / two maps: 14.7 / 24.7 us with (6, 4), 9.6 / 16.7 with (9, 1); 720p 26.4 / 47.7 -> 18.9 / 30.6
  if (a.C % (256 * v)) rx = 6, v = 4;
  a.nz = (a.C + 256 * v - 1) / (256 * v);
  a.gx = (a.w + rx - 1) / rx;
  a.total = a.gx * a.h * nmaps * a.nz;
  a.cap = (a.total + 7) / 8;
  a.xcd = cfg.dw_grid_order ? 0 : 1;
  const dim3 grid(8 * a.cap);
#define RMEM_DW_CASE(RX_, V_) \
  if (rx == RX_ && v == V_) return rmem::launch<DwArgs, dwconv5x5_split_kernel<RX_, V_>, 256>(a, grid, dim3(256), 0, s);
  RMEM_DW_CASE(9, 1) RMEM_DW_CASE(6, 4) RMEM_DW_CASE(6, 1) RMEM_DW_CASE(8, 1) RMEM_DW_CASE(12, 1) RMEM_DW_CASE(6, 2)
#undef RMEM_DW_CASE
  return RMEM_ERR_INVALID;
}

extern "C" int rmem_dwconv5x5_split2(const float* g0, const float* g1, int64_t ldg, const float* wt0, const float* wt1,
                                     int32_t h, int32_t w, int32_t C, rmem_f16* oh0, rmem_f16* ol0, rmem_f16* oh1,
                                     rmem_f16* ol1, int64_t ldo, void* stream) {
  if (!g0 || !g1 || !wt0 || !wt1 || !oh0 || !oh1 || h <= 0 || w <= 0 || (C % 4) || (ldg % 4) || (ldo % 4))
    return RMEM_ERR_INVALID;
  DwArgs a{{{g0, wt0, oh0, ol0}, {g1, wt1, oh1, ol1}}, (long)ldg, h, w, C, (long)ldo, 0, 0, 0, 0, 1};
  return launch_dwconv(a, 2, static_cast<hipStream_t>(stream));
}

extern "C" int rmem_dwconv5x5_split(const float* g, int64_t ldg, const float* wt, int32_t h, int32_t w,
                                    int32_t C, rmem_f16* oh, rmem_f16* ol, int64_t ldo, void* stream) {
  if (!g || !wt || !oh || h <= 0 || w <= 0 || (C % 4) || (ldg % 4) || (ldo % 4)) return RMEM_ERR_INVALID;
  DwArgs a{{{g, wt, oh, ol}, {g, wt, oh, ol}}, (long)ldg, h, w, C, (long)ldo, 0, 0, 0, 0, 1};
  return launch_dwconv(a, 1, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------ final GroupNorm (2 groups)
// pass A: per-block (64 tokens) double partial (sum, sumsq) for both groups
struct Gn2Args {
  const float* tgt; const float* tgt_id; int N, C; const float* gamma; const float* beta; float eps; double* ws;
  int nblk; float* out; long ldo;
  // rmem_groupnorm2_fold: split-K partials of the last projection (columns 0.. -> tgt, C.. -> tgt_id), summed into the two
  // streams in split order by the statistics pass, which writes the folded streams back (tgt / tgt_id are then outputs too)
  const float* parts; int nparts; long part_stride, ldpart;
};
// (Both passes are 27 workgroups at 480p -- a latency-bound launch each.  1024 threads per 64 tokens with every load of a
// thread issued before its first use, and the per-block partial sums reduced by a wave instead of one thread walking them:
// 12.5 + 12.3 us -> see DESIGN.md section 8.)
__device__ void gn2_stats_kernel(const Gn2Args& a, int) {
  const float* tgt = a.tgt; const float* tgt_id = a.tgt_id; const int N = a.N, C = a.C; double* ws = a.ws;
  __shared__ double red[2][2][16];
  const int t0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = C / 4, n = 64 * per;         // float4 per token / per block (C = 256: 4096 = 4 per thread)
  double s[2] = {0, 0}, q[2] = {0, 0};
  for (int i0 = tid; i0 < n; i0 += 4096) {
    float4 va[4], vb[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * 1024;
      const int tok = t0 + i / per, c = (i % per) * 4;
      ok[k] = i < n && tok < N;
      const long off = ok[k] ? (long)tok * C + c : 0;
      va[k] = *reinterpret_cast<const float4*>(tgt + off);
      vb[k] = *reinterpret_cast<const float4*>(tgt_id + off);
    }
    if (a.nparts > 0) {                        // fold the partials (fixed order: rmem_layernorm_red's arithmetic), write back
      for (int z = 0; z < a.nparts; ++z) {
        float4 pa[4], pb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * 1024;
          const int tok = t0 + i / per, c = (i % per) * 4;
          const long off = ok[k] ? (long)z * a.part_stride + (long)tok * a.ldpart + c : 0;
          pa[k] = *reinterpret_cast<const float4*>(a.parts + off);
          pb[k] = *reinterpret_cast<const float4*>(a.parts + off + (ok[k] ? C : 0));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          va[k].x += pa[k].x; va[k].y += pa[k].y; va[k].z += pa[k].z; va[k].w += pa[k].w;
          vb[k].x += pb[k].x; vb[k].y += pb[k].y; vb[k].z += pb[k].z; vb[k].w += pb[k].w;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        const int i = i0 + k * 1024;
        const long off = (long)(t0 + i / per) * C + (i % per) * 4;
        *reinterpret_cast<float4*>(const_cast<float*>(tgt) + off) = va[k];
        *reinterpret_cast<float4*>(const_cast<float*>(tgt_id) + off) = vb[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!ok[k]) continue;
      const float4 x = va[k], y = vb[k];
      s[0] += (double)x.x + (double)x.y + (double)x.z + (double)x.w;
      q[0] += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
      s[1] += (double)y.x + (double)y.y + (double)y.z + (double)y.w;
      q[1] += (double)y.x * y.x + (double)y.y * y.y + (double)y.z * y.z + (double)y.w * y.w;
    }
  }
#pragma unroll
  for (int gidx = 0; gidx < 2; ++gidx) {
    s[gidx] = wave_sum_xor_t(s[gidx]);
    q[gidx] = wave_sum_xor_t(q[gidx]);
    if (lane == 0) {
      red[gidx][0][wave] = s[gidx];
      red[gidx][1][wave] = q[gidx];
    }
  }
  __syncthreads();
  if (tid < 4) {
    const int gidx = tid >> 1, k = tid & 1;
    double t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[gidx][k][w];
    ws[(long)blockIdx.x * 4 + tid] = t;
  }
}

__device__ void gn2_apply_kernel(const Gn2Args& a, int) {
  const float* __restrict__ tgt = a.tgt; const float* __restrict__ tgt_id = a.tgt_id; const int N = a.N, C = a.C;
  const double* __restrict__ ws = a.ws;
  const float* __restrict__ gamma = a.gamma; const float* __restrict__ beta = a.beta; const float eps = a.eps; const int nblk = a.nblk;
  float* __restrict__ out = a.out; const long ldo = a.ldo;
  __shared__ float stat[4];  // mean0, rstd0, mean1, rstd1
  const int tid = threadIdx.x;
  if (tid < 64) {                              // the per-block partial sums, a strided share per lane, then a butterfly
    double p[4] = {0, 0, 0, 0};
    for (int b = tid; b < nblk; b += 64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] += ws[(long)b * 4 + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      p[j] = wave_sum_xor_t(p[j]);
    if (tid < 2) {
      const double cnt = (double)N * C;
      const double mean = p[tid * 2] / cnt;
      double var = p[tid * 2 + 1] / cnt - mean * mean;
      if (var < 0) var = 0;
      stat[tid * 2 + 0] = (float)mean;
      stat[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int t0 = blockIdx.x * 64;
  const int per = C / 4, n = 64 * per;
  for (int i0 = tid; i0 < n; i0 += 4096) {
#pragma unroll
    for (int gidx = 0; gidx < 2; ++gidx) {
      const float* __restrict__ src = gidx ? tgt_id : tgt;
      const float m = stat[gidx * 2], r = stat[gidx * 2 + 1];
      float4 v[4], g[4], b[4];
      bool ok[4];
      long offo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 1024;
        const int tok = t0 + i / per, c = (i % per) * 4;
        ok[k] = i < n && tok < N;
        v[k] = *reinterpret_cast<const float4*>(src + (ok[k] ? (long)tok * C + c : 0));
        g[k] = *reinterpret_cast<const float4*>(gamma + gidx * C + c);
        b[k] = *reinterpret_cast<const float4*>(beta + gidx * C + c);
        offo[k] = (long)tok * ldo + gidx * C + c;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        float4 o;
        o.x = (v[k].x - m) * r * g[k].x + b[k].x;
        o.y = (v[k].y - m) * r * g[k].y + b[k].y;
        o.z = (v[k].z - m) * r * g[k].z + b[k].z;
        o.w = (v[k].w - m) * r * g[k].w + b[k].w;
        *reinterpret_cast<float4*>(out + offo[k]) = o;
      }
    }
  }
}

extern "C" int rmem_groupnorm2_fold(float* tgt, float* tgt_id, const float* parts, int32_t nparts, int64_t part_stride,
                                    int64_t ldpart, int32_t N, int32_t C, const float* gamma, const float* beta, float eps,
                                    double* ws, float* out, int64_t ldo, void* stream) {
  if (!tgt || !tgt_id || !gamma || !beta || !ws || !out || N <= 0 || (C % 4) || (ldo % 4) || nparts < 0 ||
      (nparts > 0 && (!parts || (part_stride % 4) || (ldpart % 4) || ldpart < 2 * C)))
    return RMEM_ERR_INVALID;
  const int nblk = (N + 63) / 64;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Gn2Args a{tgt, tgt_id, N, C, gamma, beta, eps, ws, nblk, out, (long)ldo, parts, nparts, (long)part_stride, (long)ldpart};
  const int rc = rmem::launch<Gn2Args, gn2_stats_kernel, 1024>(a, dim3(nblk), dim3(1024), 0, s);
  if (rc != RMEM_OK) return rc;
  return rmem::launch<Gn2Args, gn2_apply_kernel, 1024>(a, dim3(nblk), dim3(1024), 0, s);
}

extern "C" int rmem_groupnorm2(const float* tgt, const float* tgt_id, int32_t N, int32_t C, const float* gamma,
                               const float* beta, float eps, double* ws, float* out, int64_t ldo, void* stream) {
  return rmem_groupnorm2_fold(const_cast<float*>(tgt), const_cast<float*>(tgt_id), nullptr, 0, 0, 0, N, C, gamma, beta, eps, ws, out,
                              ldo, stream);
}

// ------------------------------------------------------------------ ID assignment
// gather-sum over the k x k receptive field of the one-hot(+ignore) label map, then LayerNorm over C.
struct IdAssignArgs {
  const uint8_t* label; int H, W; const float* wt; const float* bias; int ncls, ksize, stride, pad, ew;
  const float* gamma; const float* beta; float eps; h16_t* oh; h16_t* ol; long ldo; float* of32; long ldof;
  int ignore_channel;
};
// block = TOK x-adjacent output tokens (1 by default), thread = one of C = 256 channels.  A token
// reads 289 weight rows of 1 KB: 484 MB per 480p launch, all L2 hits -- 25.8 us = 18.8 TB/s.  Per
// token the taps are summed in conv order.
template <int TOK, int UNR>
__device__ void id_assign_kernel(const IdAssignArgs& a, int) {
  const uint8_t* label = a.label; const int H = a.H, W = a.W; const float* wt = a.wt; const float* bias = a.bias;
  const int ncls = a.ncls, ksize = a.ksize, stride = a.stride, pad = a.pad, ew = a.ew;
  const float* gamma = a.gamma; const float* beta = a.beta; const float eps = a.eps; h16_t* oh = a.oh; h16_t* ol = a.ol;
  const long ldo = a.ldo; float* of32 = a.of32; const long ldof = a.ldof; const int ignore_channel = a.ignore_channel;
  __shared__ float red[TOK][4];
  __shared__ float red2[TOK][4];
  __shared__ int cls_s[TOK][1024];     // weight-row offset per tap of the receptive field (-1 = no channel)
  const int nbx = (ew + TOK - 1) / TOK;
  const int oy = blockIdx.x / nbx, ox0 = (blockIdx.x - oy * nbx) * TOK;
  const int c = threadIdx.x;
  const int lane = c & 63, wave = c >> 6;
  const int ntap = ksize * ksize;
  const int ntap_pad = (ntap + UNR - 1) / UNR * UNR;       // <= 1024: ksize <= 31 (checked by the launcher)
  for (int i = c; i < TOK * ntap_pad; i += 256) {
    const int j = i / ntap_pad, t = i - j * ntap_pad;
    if (t >= ntap) {
      cls_s[j][t] = -1;
      continue;
    }
    const int dy = t / ksize, dx = t - dy * ksize;
    const int y = oy * stride - pad + dy, x = (ox0 + j) * stride - pad + dx;
    int cls = -1;
    if (ox0 + j < ew && y >= 0 && y < H && x >= 0 && x < W) {
      cls = label[(long)y * W + x];
      if (cls == 255) cls = ignore_channel ? ncls - 1 : -1;   // ignore channel is the last one; reference frames carry none
      else if (cls >= ncls - 1) cls = -1;  // ids above max_obj have no one-hot channel
    }
    cls_s[j][t] = cls >= 0 ? (cls * ntap + t) * 256 : -1;      // element offset of the tap's weight row
  }
  __syncthreads();
  float acc[TOK];
#pragma unroll
  for (int j = 0; j < TOK; ++j) acc[j] = bias[c];
  // UNR taps per step: their row offsets move to SGPRs (block-uniform: the address arithmetic runs
  // on the scalar unit -- as per-lane 64-bit multiplies it was the bound of this kernel), the UNR loads
  // are unconditional (row 0 stands in for "no channel", zeroed by a select) and in flight together;
  // the sums stay in conv order.
  for (int t0 = 0; t0 < ntap; t0 += UNR) {
#pragma unroll
    for (int j = 0; j < TOK; ++j) {
      int off[UNR];
      float wv[UNR];
#pragma unroll
      for (int i = 0; i < UNR; ++i) off[i] = __builtin_amdgcn_readfirstlane(cls_s[j][t0 + i]);
#pragma unroll
      for (int i = 0; i < UNR; ++i) wv[i] = (wt + (off[i] > 0 ? off[i] : 0))[c];
#pragma unroll
      for (int i = 0; i < UNR; ++i) acc[j] += off[i] >= 0 ? wv[i] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < TOK; ++j) {
    float s = acc[j];
    s = wave_sum_xor_t(s);
    if (lane == 0) red[j][wave] = s;
  }
  __syncthreads();
  float d[TOK];
#pragma unroll
  for (int j = 0; j < TOK; ++j) {
    const float mean = (red[j][0] + red[j][1] + red[j][2] + red[j][3]) * (1.0f / 256.0f);
    d[j] = acc[j] - mean;
    float ss = d[j] * d[j];
    ss = wave_sum_xor_t(ss);
    if (lane == 0) red2[j][wave] = ss;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TOK; ++j) {
    if (ox0 + j >= ew) continue;
    const long tok = (long)oy * ew + ox0 + j;
    float yv = acc[j];
    if (gamma) {
      const float var = (red2[j][0] + red2[j][1] + red2[j][2] + red2[j][3]) * (1.0f / 256.0f);
      yv = d[j] / sqrtf(var + eps) * gamma[c] + beta[c];
    }
    if (of32) of32[tok * ldof + c] = yv;
    if (oh) {
      h16_t hi, lo;
      split_f16(yv, hi, lo);
      oh[tok * ldo + c] = hi;
      if (ol) ol[tok * ldo + c] = lo;
    }
  }
}

extern "C" int rmem_id_assign(const uint8_t* label, int32_t H, int32_t W, const float* wt, const float* bias,
                              int32_t ncls, int32_t ksize, int32_t stride, int32_t pad, int32_t eh, int32_t ew,
                              int32_t C, const float* gamma, const float* beta, float eps, rmem_f16* oh,
                              rmem_f16* ol, int64_t ldo, float* of32, int64_t ldof, int32_t ignore_channel,
                              void* stream) {
  if (!label || !wt || !bias || C != 256 || eh <= 0 || ew <= 0 || ncls < 2 || ksize <= 0 || ksize > 31)
    return RMEM_ERR_INVALID;
  IdAssignArgs a{label, H, W, wt, bias, ncls, ksize, stride, pad, ew, gamma, beta, eps, oh, ol, (long)ldo, of32,
                 (long)ldof, ignore_channel};
  const int tok = rmem_config().ida_tokens, unr = rmem_config().ida_unroll;   // (1, 16): 480p 25.8 us; 2 tokens per block 30.6
  hipStream_t s = static_cast<hipStream_t>(stream);
#define RMEM_IDA_CASE(T_, U_) \
  if (tok == T_ && unr == U_) \
    return rmem::launch<IdAssignArgs, id_assign_kernel<T_, U_>, 256>(a, dim3(eh * ((ew + T_ - 1) / T_)), dim3(256), 0, s);
  RMEM_IDA_CASE(1, 16) RMEM_IDA_CASE(1, 8) RMEM_IDA_CASE(1, 32) RMEM_IDA_CASE(2, 16)
#undef RMEM_IDA_CASE
  return RMEM_ERR_INVALID;
}

// ------------------------------------------------------------------ RMem relevance reduce
struct MassReduceArgs {
  const float* mass; int N, T; const float* fg; float* out;
};
__device__ void mass_reduce_kernel(const MassReduceArgs& a, int) {
  const float* mass = a.mass; const int N = a.N, T = a.T; const float* fg = a.fg; float* out = a.out;
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = 0; t < T; ++t) {
    float s = 0.f;
    for (int q = tid; q < N; q += 1024) s += mass[(long)q * T + t] * fg[q];
    s = wave_sum_xor_t(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
      float r = 0.f;
      for (int i = 0; i < 16; ++i) r += red[i];
      out[t] = r;
    }
    __syncthreads();
  }
}

extern "C" int rmem_attn_mass_reduce(const float* mass, int32_t N, int32_t T, const float* fg, float* out,
                                     void* stream) {
  if (!mass || !fg || !out || N <= 0 || T <= 0) return RMEM_ERR_INVALID;
  MassReduceArgs a{mass, N, T, fg, out};
  return rmem::launch<MassReduceArgs, mass_reduce_kernel, 1024>(a, dim3(1), dim3(1024), 0,
                                                                 static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------ RMem eviction on the device
// restrict_long_memories (layers/transformer.py:880-991) without a host round trip: foreground weights,
// EMA(0.8) of the normalised attention mass, visit counts, UCB bonus 1.5*sqrt(log(sum c)/(c+8)), argmin, and
// the deletion of the dropped slot from the logical->physical map -- all in device memory (one thread: the bank
// holds <= 16 slots).  The arithmetic is the reference's, operation by operation, in fp32 with every product and
// sum rounded on its own (no fma contraction); the sum that normalises the mass is numpy's pairwise order for
// < 16 values (what the host rule rmem_amd.lstt.rmem_policy_step uses); log of the (integer) visit total is
// taken in double and rounded once.

// fg[q] = 1 - softmax_c(bilinear_align_corners(logits -> h x w))[0]   (engines/aot_engine.py:350-356); the
// interpolation arithmetic of torch's upsample_bilinear2d kernel (fp32 source index = scale * dst)
__global__ __launch_bounds__(256) void fg_weights_kernel(const float* __restrict__ lg, int C, int Hl, int Wl, int h, int w,
                                                        float rh, float rw, float* __restrict__ fg) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= h * w) return;
  const int Y = q / w, X = q - Y * w;
  const float fy = rh * (float)Y, fx = rw * (float)X;
  const int y0 = (int)fy, x0 = (int)fx;
  const int yp = (y0 < Hl - 1) ? 1 : 0, xp = (x0 < Wl - 1) ? 1 : 0;
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  float v[16];
  float mx = -3.0e38f;
  for (int c = 0; c < C; ++c) {
    const float* p = lg + ((long)c * Hl + y0) * Wl + x0;
    v[c] = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[(long)yp * Wl] + lx1 * p[(long)yp * Wl + xp]);
    mx = fmaxf(mx, v[c]);
  }
  float sum = 0.f, e0 = 0.f;
  for (int c = 0; c < C; ++c) {
    const float e = expf(v[c] - mx);
    if (c == 0) e0 = e;
    sum += e;
  }
  fg[q] = 1.f - e0 / sum;
}

extern "C" int rmem_fg_weights(const float* logits, int32_t C, int32_t Hl, int32_t Wl, int32_t h, int32_t w, float* fg,
                               void* stream) {
  if (!logits || !fg || C <= 0 || C > 16 || Hl <= 0 || Wl <= 0 || h <= 0 || w <= 0) return RMEM_ERR_INVALID;
  const float rh = h > 1 ? (float)(Hl - 1) / (float)(h - 1) : 0.f, rw = w > 1 ? (float)(Wl - 1) / (float)(w - 1) : 0.f;
  hipLaunchKernelGGL(fg_weights_kernel, dim3((h * w + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), logits, C,
                     Hl, Wl, h, w, rh, rw, fg);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

__global__ void bank_edit_kernel(int32_t* maps, rmem_bank_state* st, int slot, int frame_index, int reset) {
  if (threadIdx.x != 0) return;
  if (reset) {                                       // init_memory (transformer.py:993-998): the bank is this one frame
    st->T = 0;
    st->last_drop = -1;
  }
  const int T = st->T;
  if (T >= 16) return;
  maps[T] = slot;
  st->index[T] = frame_index;
  st->visits[T] = 0;                                 // "not in stored_frame_times"
  st->has_ema[T] = 0;                                // "not in stored_attn_weight_dict"
  st->ema[T] = 0.f;
  st->T = T + 1;
}

extern "C" int rmem_bank_reset(int32_t* maps, rmem_bank_state* st, int32_t slot, int32_t frame_index, void* stream) {
  if (!maps || !st || slot < 0) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(bank_edit_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), maps, st, slot, frame_index, 1);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_bank_append(int32_t* maps, rmem_bank_state* st, int32_t slot, int32_t frame_index, void* stream) {
  if (!maps || !st || slot < 0) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(bank_edit_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), maps, st, slot, frame_index, 0);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

__device__ __forceinline__ float sum_np_order(const float* a, int n) {      // numpy's float32 add.reduce for n < 128
  if (n < 8) {
    float r = 0.f;
    for (int i = 0; i < n; ++i) r = __fadd_rn(r, a[i]);
    return r;
  }
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i + 8 <= n; i += 8)
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, a[i]);
  return res;
}

__global__ void bank_policy_kernel(int32_t* maps, rmem_bank_state* st, const float* __restrict__ w_raw, int n_att, int cap,
                                   int former, int32_t* result) {
  if (threadIdx.x != 0) return;
  const int T = st->T;                               // includes the slot appended by this update (not attended)
  if (n_att > T) n_att = T;
  float w[16], c[16];
  for (int i = 0; i < n_att; ++i) w[i] = w_raw[i];
  const float wsum = sum_np_order(w, n_att);
  for (int i = 0; i < n_att; ++i) {                  // normalise, moving mean over the slots seen before (:905-928)
    float x = __fdiv_rn(w[i], wsum);
    if (st->has_ema[i]) x = __fadd_rn(__fmul_rn(0.2f, st->ema[i]), __fmul_rn(0.8f, x));
    w[i] = x;
    st->ema[i] = x;
    st->has_ema[i] = 1;
  }
  for (int i = n_att; i < T; ++i) st->has_ema[i] = 0;     // the dictionary is rebuilt from the attended slots only
  for (int i = 0; i < T; ++i) st->visits[i] += 1;         // (:930-941; 0 = "not stored" -> 1)
  int drop = former;
  const int nc = T - 1;                              // counts over indexes[:-1]
  if (nc > 0) {
    for (int i = 0; i < nc; ++i) c[i] = (float)st->visits[i];
    c[0] = (float)nc;
    const float lg = (float)log((double)sum_np_order(c, nc));
    if (n_att > 1) {
      float best = 0.f;
      for (int i = 1; i < n_att; ++i) {
        const float score = __fadd_rn(w[i], __fmul_rn(1.5f, __fsqrt_rn(__fdiv_rn(lg, __fadd_rn(c[i], 8.0f)))));
        if (i == 1 || score < best) {
          best = score;
          drop = i;
        }
      }
    }
  }
  int dropped = -1;
  if (T > cap && drop < T) {                         // (:967-991) delete the slot in every layer = delete its map entry
    dropped = drop;
    for (int i = drop; i + 1 < T; ++i) {
      maps[i] = maps[i + 1];
      st->index[i] = st->index[i + 1];
      st->visits[i] = st->visits[i + 1];
      st->has_ema[i] = st->has_ema[i + 1];
      st->ema[i] = st->ema[i + 1];
    }
    st->T = T - 1;
  }
  st->last_drop = dropped;
  st->steps += 1;
  if (result) {
    result[1] = dropped;
    result[2] = st->T;
    __threadfence_system();
    result[0] = st->steps;                           // sequence number last: a host that sees it sees the rest
  }
}

extern "C" int rmem_bank_policy_step(int32_t* maps, rmem_bank_state* st, const float* w, int32_t n_att, int32_t cap,
                                     int32_t former, int32_t* result, void* stream) {
  if (!maps || !st || !w || n_att < 0 || n_att > 16 || cap <= 0) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(bank_policy_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), maps, st, w, n_att, cap,
                     former, result);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ fp32 -> planes
__global__ void split_planes_kernel(const float* x, long n, h16_t* hi, h16_t* lo) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    h16_t h, l;
    split_f16(x[i], h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
  }
}

extern "C" int rmem_split_planes(const float* x, int64_t n, rmem_f16* hi, rmem_f16* lo, void* stream) {
  if (!x || !hi || n <= 0) return RMEM_ERR_INVALID;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, (long)n, hi, lo);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}


// ------------------------------------------------------------------ GroupNorm on NCHW (+ReLU)
// Support kernel for the FPN head (outside the LSTT): PyTorch's GroupNorm forward runs one
// workgroup per (image, group) -- 8 workgroups at batch 1 -- and costs ~100 us per call on
// MI355X.  Here a group (a contiguous segment of (C/G)*HW floats) is reduced by `ns` blocks
// into double partials, and the apply pass fuses the affine and the ReLU.
__global__ __launch_bounds__(256) void gn_nchw_stats_kernel(const float* x, const float* cbias, int cpg, long hw, long seg,
                                                            int ns, double* ws) {
  __shared__ double red[2][4];
  const int g = blockIdx.y, sidx = blockIdx.x;
  const long per = ((seg / 4 + ns - 1) / ns) * 4;
  long lo = (long)sidx * per, hi = lo + per;
  if (hi > seg) hi = seg;
  const float* xb = x + (long)g * seg;
  double s = 0, q = 0;
  // cbias (may be NULL): the producing convolution's per-channel bias, added here in fp32 exactly
  // as the separate bias pass of the convolution would (x + b rounded once)
  const float* cb = cbias ? cbias + (long)g * cpg : nullptr;
  for (long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
    if (i + 3 < hi) {
      float4 v = *reinterpret_cast<const float4*>(xb + i);
      if (cb) {
        v.x += cb[i / hw]; v.y += cb[(i + 1) / hw]; v.z += cb[(i + 2) / hw]; v.w += cb[(i + 3) / hw];
      }
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (long j = i; j < hi; ++j) {
        const float v = cb ? xb[j] + cb[j / hw] : xb[j];
        s += v;
        q += (double)v * v;
      }
    }
  }
  s = wave_sum_xor_t(s);
  q = wave_sum_xor_t(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x < 2)
    ws[((long)g * ns + sidx) * 2 + threadIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void gn_nchw_apply_kernel(const float* x, const float* cbias, float* y, long seg, int cpg, long hw,
                                                            int ns, const double* ws, const float* gamma,
                                                            const float* beta, float eps, int relu) {
  __shared__ float stat[2];
  const int g = blockIdx.y;
  if (threadIdx.x == 0) {
    double s = 0, q = 0;
    for (int i = 0; i < ns; ++i) {
      s += ws[((long)g * ns + i) * 2];
      q += ws[((long)g * ns + i) * 2 + 1];
    }
    const double mean = s / (double)seg;
    double var = q / (double)seg - mean * mean;
    if (var < 0) var = 0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  const float* xb = x + (long)g * seg;
  float* yb = y + (long)g * seg;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < seg; i += (long)gridDim.x * 1024) {
    float v[4];
    const bool full = i + 3 < seg;
    if (full) {
      const float4 t = *reinterpret_cast<const float4*>(xb + i);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      for (int e = 0; e < 4; ++e) v[e] = (i + e < seg) ? xb[i + e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = g * cpg + (int)((i + e) / hw);
      const int cc = c < (g + 1) * cpg ? c : (g + 1) * cpg - 1;
      const float xv = cbias ? v[e] + cbias[cc] : v[e];
      float o = (xv - mean) * rstd * gamma[cc] + beta[cc];
      if (relu) o = o > 0.f ? o : 0.f;
      v[e] = o;
    }
    if (full) {
      *reinterpret_cast<float4*>(yb + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int e = 0; e < 4; ++e)
        if (i + e < seg) yb[i + e] = v[e];
    }
  }
}

static int groupnorm_nchw_impl(const float* x, const float* cbias, float* y, int32_t C, int64_t HW, int32_t groups,
                               const float* gamma, const float* beta, float eps, int32_t relu, double* ws,
                               void* stream) {
  if (!x || !y || !gamma || !beta || !ws || C <= 0 || groups <= 0 || (C % groups) || HW <= 0) return RMEM_ERR_INVALID;
  const int cpg = C / groups;
  const long seg = (long)cpg * HW;
  if ((seg % 4) != 0) return RMEM_ERR_INVALID;   // keeps every group segment 16-byte aligned
  const int ns = 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gn_nchw_stats_kernel, dim3(ns, groups), dim3(256), 0, s, x, cbias, cpg, (long)HW, seg, ns, ws);
  long ab = (seg / 4 + 255) / 256;
  if (ab > 64) ab = 64;
  hipLaunchKernelGGL(gn_nchw_apply_kernel, dim3((unsigned)ab, groups), dim3(256), 0, s, x, cbias, y, seg, cpg, (long)HW,
                     ns, ws, gamma, beta, eps, relu);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_groupnorm_nchw(const float* x, float* y, int32_t C, int64_t HW, int32_t groups,
                                   const float* gamma, const float* beta, float eps, int32_t relu, double* ws,
                                   void* stream) {
  return groupnorm_nchw_impl(x, nullptr, y, C, HW, groups, gamma, beta, eps, relu, ws, stream);
}

extern "C" int rmem_groupnorm_nchw_bias(const float* x, const float* conv_bias, float* y, int32_t C, int64_t HW,
                                        int32_t groups, const float* gamma, const float* beta, float eps,
                                        int32_t relu, double* ws, void* stream) {
  if (!conv_bias) return RMEM_ERR_INVALID;
  return groupnorm_nchw_impl(x, conv_bias, y, C, HW, groups, gamma, beta, eps, relu, ws, stream);
}

// ------------------------------------------------------------------ planes transpose
__global__ __launch_bounds__(256) void transpose_planes_kernel(const h16_t* ih, const h16_t* il, long ld, int N,
                                                               int C, h16_t* oh, h16_t* ol, long ldo) {
  __shared__ h16_t th[64][66];
  __shared__ h16_t tl[64][66];
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    const bool ok = n0 + r < N && c0 + c < C;
    th[r][c] = ok ? ih[(long)(n0 + r) * ld + c0 + c] : (h16_t)0;
    if (il) tl[r][c] = ok ? il[(long)(n0 + r) * ld + c0 + c] : (h16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && n0 + r < N) {
      oh[(long)(c0 + c) * ldo + n0 + r] = th[r][c];
      if (ol) ol[(long)(c0 + c) * ldo + n0 + r] = tl[r][c];
    }
  }
}

extern "C" int rmem_transpose_planes(const rmem_f16* ih, const rmem_f16* il, int64_t ld, int32_t N, int32_t C,
                                     rmem_f16* oh, rmem_f16* ol, int64_t ldo, void* stream) {
  if (!ih || !oh || N <= 0 || C <= 0) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(transpose_planes_kernel, dim3((N + 63) / 64, (C + 63) / 64), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ih, il, (long)ld, N, C, oh, ol, (long)ldo);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ a + b -> fp32 / planes
__global__ void add_split_kernel(const float* a, const float* b, long n, float* dst, h16_t* oh, h16_t* ol) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = a[i] + (b ? b[i] : 0.f);
    if (dst) dst[i] = v;
    if (oh) {
      h16_t h, l;
      split_f16(v, h, l);
      oh[i] = h;
      if (ol) ol[i] = l;
    }
  }
}

// up to eight such sums of n elements each in ONE launch (blockIdx.y = problem)
struct AddMultiArgs {
  rmem_add_args p[8];
  long n;
};
__global__ void add_split_multi_kernel(AddMultiArgs g) {
  const rmem_add_args& q = g.p[blockIdx.y];
  const float* a = q.a;
  const float* b = q.b;
  float* dst = q.dst;
  h16_t* oh = reinterpret_cast<h16_t*>(q.oh);
  h16_t* ol = reinterpret_cast<h16_t*>(q.ol);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += (long)gridDim.x * blockDim.x) {
    const float v = a[i] + (b ? b[i] : 0.f);
    if (dst) dst[i] = v;
    if (oh) {
      h16_t h, l;
      split_f16(v, h, l);
      oh[i] = h;
      if (ol) ol[i] = l;
    }
  }
}

extern "C" int rmem_add_split_multi(const rmem_add_args* p, int32_t n, int64_t nelem, void* stream) {
  if (!p || n <= 0 || n > 8 || nelem <= 0) return RMEM_ERR_INVALID;
  AddMultiArgs g;
  for (int i = 0; i < n; ++i) {
    if (!p[i].a || (!p[i].dst && !p[i].oh)) return RMEM_ERR_INVALID;
    g.p[i] = p[i];
  }
  g.n = (long)nelem;
  long blocks = (nelem + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(add_split_multi_kernel, dim3((unsigned)blocks, n), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_add_split(const float* a, const float* b, int64_t n, float* dst, rmem_f16* oh, rmem_f16* ol,
                              void* stream) {
  if (!a || n <= 0 || (!dst && !oh)) return RMEM_ERR_INVALID;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(add_split_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b,
                     (long)n, dst, oh, ol);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ GroupNorm over tokens + GELU
// x token-major [N][C]; group g = channels [g*cpg, (g+1)*cpg) of every token.
__global__ __launch_bounds__(256) void gn_tok_stats_kernel(const float* x, int N, int C, int cpg, int ns,
                                                           double* ws) {
  __shared__ double red[2][4];
  const int g = blockIdx.y, sidx = blockIdx.x;
  const int per = (N + ns - 1) / ns;
  const int t0 = sidx * per;
  int t1 = t0 + per;
  if (t1 > N) t1 = N;
  double s = 0, q = 0;
  const int tot = (t1 - t0) * cpg;
  for (int i = threadIdx.x; i < tot; i += 256) {
    const int tok = t0 + i / cpg, c = g * cpg + i % cpg;
    const float v = x[(long)tok * C + c];
    s += v;
    q += (double)v * v;
  }
  s = wave_sum_xor_t(s);
  q = wave_sum_xor_t(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x < 2)
    ws[((long)g * ns + sidx) * 2 + threadIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void gn_tok_gelu_kernel(const float* x, int N, int C, int cpg, int ns,
                                                          const double* ws, const float* gamma, const float* beta,
                                                          float eps, float* y) {
  extern __shared__ float stat[];   // [groups][2]
  const int groups = C / cpg;
  for (int g = threadIdx.x; g < groups; g += 256) {
    double s = 0, q = 0;
    for (int i = 0; i < ns; ++i) {
      s += ws[((long)g * ns + i) * 2];
      q += ws[((long)g * ns + i) * 2 + 1];
    }
    const double cnt = (double)N * cpg;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0) var = 0;
    stat[g * 2] = (float)mean;
    stat[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const long n4 = (long)N * C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c = (int)((i * 4) % C);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c);
    const int g = c / cpg;              // cpg % 4 == 0: the 4 channels share a group
    const float m = stat[g * 2], r = stat[g * 2 + 1];
    float o[4] = {(v.x - m) * r * gm.x + bt.x, (v.y - m) * r * gm.y + bt.y, (v.z - m) * r * gm.z + bt.z,
                  (v.w - m) * r * gm.w + bt.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.0f + erff(o[e] * 0.70710678118654752440f));
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int rmem_gn_gelu_tokens(const float* x, int32_t N, int32_t C, int32_t groups, const float* gamma,
                                   const float* beta, float eps, double* ws, float* y, void* stream) {
  if (!x || !y || !ws || !gamma || !beta || N <= 0 || groups <= 0 || (C % groups) || ((C / groups) % 4))
    return RMEM_ERR_INVALID;
  const int cpg = C / groups, ns = 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gn_tok_stats_kernel, dim3(ns, groups), dim3(256), 0, s, x, N, C, cpg, ns, ws);
  long blocks = ((long)N * C / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gn_tok_gelu_kernel, dim3((unsigned)blocks), dim3(256), groups * 2 * sizeof(float), s, x, N, C,
                     cpg, ns, ws, gamma, beta, eps, y);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ temporal-PE bias per head
struct PeRowsH { int row[16]; };
__global__ void pe_bias_heads_kernel(const float* Q, long ldq, const float* cur_pe, const float* mem_pe,
                                     PeRowsH rows, int T, int N, int heads, float* bias) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= N) return;
  const int d = heads * 32;
  for (int t = 0; t < T; ++t) {
    for (int c0 = 0; c0 < d; c0 += 64) {      // lanes 0-31 -> head c0/32, lanes 32-63 -> head c0/32 + 1
      const int c = c0 + lane;
      float s = (Q[(long)q * ldq + c] + cur_pe[c]) * mem_pe[(long)rows.row[t] * d + c];
      s = wave_sum_xor_t<16>(s);
      if ((lane & 31) == 0) bias[((long)q * heads + (c >> 5)) * T + t] = s;
    }
  }
}

extern "C" int rmem_pe_bias_heads(const float* Q, int64_t ldq, const float* cur_pe, const float* mem_pe,
                                  const int32_t* pe_row_host, int32_t T, int32_t N, int32_t heads, float* bias,
                                  void* stream) {
  if (!Q || !cur_pe || !mem_pe || !pe_row_host || !bias || T <= 0 || T > 16 || N <= 0 || heads <= 0 || (heads % 2))
    return RMEM_ERR_INVALID;
  PeRowsH rows;
  for (int t = 0; t < 16; ++t) rows.row[t] = t < T ? pe_row_host[t] : 0;
  hipLaunchKernelGGL(pe_bias_heads_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), Q,
                     (long)ldq, cur_pe, mem_pe, rows, T, N, heads, bias);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ conv epilogue (encoder support)
// y = act(x + bias[c] (+ residual)) in place on a contiguous NCHW batch-1 tensor: replaces the
// separate bias-add / residual-add / ReLU launches PyTorch-ROCm issues after every MIOpen conv.
__global__ __launch_bounds__(256) void bias_act_nchw_kernel(float* x, const float* bias, const float* res, unsigned hw,
                                                            unsigned C, unsigned n, int relu) {
  // 32-bit indices (n < 2^32 checked by the host): one division per float4 instead of four 64-bit
  // ones -- the integer divisions were most of this kernel's instructions
  for (unsigned i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < n; i += gridDim.x * 1024u) {
    const unsigned p0 = i / hw;                    // (image, channel) plane index
    const unsigned left = (p0 + 1) * hw - i;       // elements of plane p0 from i on
    const unsigned c0 = p0 % C, c1 = (p0 + 1) % C;
    if (i + 3 < n) {
      float4 v = *reinterpret_cast<const float4*>(x + i);
      float o[4] = {v.x, v.y, v.z, v.w};
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      if (res) {
        const float4 t = *reinterpret_cast<const float4*>(res + i);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
      }
      const float b0 = bias[c0];
      const float b1 = left < 4 ? bias[c1] : b0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = o[e] + ((unsigned)e < left ? b0 : b1) + r[e];
        o[e] = relu ? fmaxf(t, 0.f) : t;
      }
      *reinterpret_cast<float4*>(x + i) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      for (unsigned j = i; j < n; ++j) {
        float t = x[j] + bias[(j / hw) % C] + (res ? res[j] : 0.f);
        x[j] = relu ? fmaxf(t, 0.f) : t;
      }
    }
  }
}

static int bias_act_impl(float* x, const float* bias, const float* residual, int32_t B, int32_t C, int64_t HW,
                         int32_t relu, void* stream) {
  if (!x || !bias || B <= 0 || C <= 0 || HW <= 0) return RMEM_ERR_INVALID;
  const long n = (long)B * C * HW;
  if (n >= (1l << 32) - 4096 || HW < 4) return RMEM_ERR_INVALID;   // 32-bit indexing; a float4 spans <= 2 channels
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (residual && (reinterpret_cast<uintptr_t>(residual) & 15)))
    return RMEM_ERR_INVALID;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bias_act_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     bias, residual, (unsigned)HW, (unsigned)C, (unsigned)n, relu);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_bias_act_nchw(float* x, const float* bias, const float* residual, int32_t C, int64_t HW,
                                  int32_t relu, void* stream) {
  return bias_act_impl(x, bias, residual, 1, C, HW, relu, stream);
}

extern "C" int rmem_bias_act_nchw_batched(float* x, const float* bias, const float* residual, int32_t B, int32_t C,
                                          int64_t HW, int32_t relu, void* stream) {
  return bias_act_impl(x, bias, residual, B, C, HW, relu, stream);
}

// ------------------------------------------------------------------ FPN skip merge
// y[c][Y][X] = (y[c][Y][X] + bias[c]) + bilinear(x[c] -> H x W)[Y][X]   in place: the
// "adapter(shortcut) + upsample(x)" of the FPN head (decoders/fpn.py:53-60) in one pass over
// the high-resolution map instead of bias-add, F.interpolate and add (three passes, the
// generic upsample kernel alone runs at 0.3 TB/s).  Same interpolation arithmetic as
// torch's upsample_bilinear2d (area_pixel_compute_source_index, fp32).
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* yin, float* y, const float* __restrict__ bias,
                                                          const float* __restrict__ x, int H, int W, int h, int w,
                                                          float rh, float rw, int align) {
  const int X = blockIdx.x * 256 + threadIdx.x;
  const int cy = blockIdx.y;
  const int c = cy / H, Y = cy - c * H;
  if (X >= W) return;
  float fy, fx;
  if (align) {
    fy = rh * (float)Y;
    fx = rw * (float)X;
  } else {
    fy = fmaxf(rh * ((float)Y + 0.5f) - 0.5f, 0.f);
    fx = fmaxf(rw * ((float)X + 0.5f) - 0.5f, 0.f);
  }
  const int y0 = (int)fy, x0 = (int)fx;
  const int yp = (y0 < h - 1) ? 1 : 0, xp = (x0 < w - 1) ? 1 : 0;
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  const float* q = x + ((long)c * h + y0) * w + x0;
  const float up = ly0 * (lx0 * q[0] + lx1 * q[xp]) + ly1 * (lx0 * q[(long)yp * w] + lx1 * q[(long)yp * w + xp]);
  const long oi = ((long)c * H + Y) * W + X;
  const float b = bias ? bias[c] : 0.f;
  y[oi] = (yin[oi] + b) + up;
}

static int upsample_add_impl(const float* yin, float* y, const float* bias, const float* x, int32_t C, int32_t H,
                             int32_t W, int32_t h, int32_t w, int32_t align_corners, void* stream) {
  if (!y || !yin || !x || C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return RMEM_ERR_INVALID;
  if ((long)C * H > 65535) return RMEM_ERR_INVALID;
  float rh, rw;
  if (align_corners) {
    rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  } else {
    rh = (float)h / (float)H;
    rw = (float)w / (float)W;
  }
  hipLaunchKernelGGL(upsample_add_kernel, dim3((W + 255) / 256, C * H), dim3(256), 0, static_cast<hipStream_t>(stream),
                     yin, y, bias, x, H, W, h, w, rh, rw, align_corners);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_upsample_add_nchw(float* y, const float* bias, const float* x, int32_t C, int32_t H, int32_t W,
                                      int32_t h, int32_t w, int32_t align_corners, void* stream) {
  return upsample_add_impl(y, y, bias, x, C, H, W, h, w, align_corners, stream);
}

extern "C" int rmem_upsample_add_nchw_out(const float* y_in, float* y_out, const float* bias, const float* x,
                                          int32_t C, int32_t H, int32_t W, int32_t h, int32_t w,
                                          int32_t align_corners, void* stream) {
  return upsample_add_impl(y_in, y_out, bias, x, C, H, W, h, w, align_corners, stream);
}

// ------------------------------------------------------------------ slot map publish
// The logical->physical slot map is tiny host state; writing it with a kernel whose payload
// travels in the kernel arguments keeps the update stream-ordered WITHOUT the host-blocking
// pageable H2D copy (which serialised host and GPU once per frame).
struct IntPayload { int v[32]; };
__global__ void set_ints_kernel(int* dst, IntPayload p, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}

extern "C" int rmem_set_ints(int32_t* dst, const int32_t* host_vals, int32_t n, void* stream) {
  if (!dst || !host_vals || n <= 0 || n > 32) return RMEM_ERR_INVALID;
  IntPayload p;
  for (int i = 0; i < 32; ++i) p.v[i] = i < n ? host_vals[i] : 0;
  hipLaunchKernelGGL(set_ints_kernel, dim3(1), dim3(32), 0, static_cast<hipStream_t>(stream), dst, p, n);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_set_host_wait(int32_t device, int32_t blocking) {
  int n = 0, prev = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return RMEM_ERR_INVALID;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  if (hipSetDevice(device) != hipSuccess) return RMEM_ERR_INVALID;
  const hipError_t e = hipSetDeviceFlags(blocking ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
  if (had && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return RMEM_ERR_LAUNCH;
  }
  return RMEM_OK;
}

extern "C" int rmem_abi_version(void) { return 18; }   // 18: rmem_layernorm_multi, rmem_add_split_multi (several LayerNorms / sums in one launch); 17: rmem_configure (no getenv in the library), rmem_read_args.gate / gout (single-split reads gate their own output), rmem_ln_linear_grouped removed; 16: rmem_ln_linear_grouped (LayerNorm + grouped projections, row tile resident in LDS); 15: rmem_set_host_wait; 14: rmem_read_args.nfull / pf (uneven key splits), rmem_layernorm_cn; 13: streaming projection kernel (rmem_linear_args.tile 0 / 256), rmem_linear_trace; 12: rmem_read_args.sched (unit queue of the paired read); 11: read64 kernel (ncols = 1024), rmem_attn_read_trace, device-side eviction (rmem_fg_weights, rmem_bank_*); 10: the three-launch materialised-P attention (rmem_attn_scores[2] / pv / combine[2]) removed; 9: launch recorder (rmem_rec_*, rmem_launch_recorded); 8: rmem_f16 naming, rmem_id_assign(ignore_channel), fused memory read
