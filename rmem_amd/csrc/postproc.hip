// postproc.hip -- the clip driver's per-frame mask post-processing on device:
//   decoder logits [C, h, w] (one per test-time augmentation)
//     -> bilinear upsample to the original frame size (engines/aot_engine.py:457-463)
//     -> un-flip, softmax over C, mean over augmentations, argmax
//        (managers/evaluator.py:424-441)
//     -> uint8 label map [H0, W0]
//   and the nearest-neighbour resize (+ flip) of that label map to an engine's input size
//   that feeds update_memory (managers/evaluator.py:506-523).
// Byte/HBM-bound work: one thread per output pixel, logits stay L2-resident (1.1 MB at 480p),
// 0.4 MB of labels written instead of an 18 MB fp32 probability volume.
#include "rmem_common.h"
#include "../../include/rmem_hip.h"

namespace {

constexpr int MAXC = 16;
constexpr int MAXSRC = 8;

struct LabelSrcs {
  const float* logits[MAXSRC];
  int h[MAXSRC], w[MAXSRC], flip[MAXSRC];
  float rh[MAXSRC], rw[MAXSRC];
};

// torch's area_pixel_compute_source_index (bilinear)
__device__ __forceinline__ float src_index(float scale, int dst, int align) {
  if (align) return scale * (float)dst;
  const float s = scale * ((float)dst + 0.5f) - 0.5f;
  return s < 0.f ? 0.f : s;
}

template <int NSRC_IS_ONE>
__global__ void __launch_bounds__(256) labels_kernel(LabelSrcs S, int nsrc, int C, int align, int H0, int W0,
                                                     uint8_t* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W0 || y >= H0) return;
  float acc[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
  for (int s = 0; s < nsrc; ++s) {
    const int h = S.h[s], w = S.w[s];
    const int xx = S.flip[s] ? (W0 - 1 - x) : x;
    const float fy = src_index(S.rh[s], y, align), fx = src_index(S.rw[s], xx, align);
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = (y0 < h - 1) ? 1 : 0, xp = (x0 < w - 1) ? 1 : 0;
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    const float* p = S.logits[s] + (long)y0 * w + x0;
    const long hw = (long)h * w;
    float v[MAXC];
    float m = -3.0e38f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < C) {
        const float* q = p + c * hw;
        v[c] = ly0 * (lx0 * q[0] + lx1 * q[xp]) + ly1 * (lx0 * q[(long)yp * w] + lx1 * q[(long)yp * w + xp]);
        m = fmaxf(m, v[c]);
      }
    }
    if (NSRC_IS_ONE) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) acc[c] = v[c];
    } else {
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) {
          v[c] = expf(v[c] - m);
          sum += v[c];
        }
      const float inv = 1.f / sum;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) acc[c] += v[c] * inv;
    }
  }
  int best = 0;
  float bv = acc[0];
#pragma unroll
  for (int c = 1; c < MAXC; ++c)
    if (c < C && acc[c] > bv) {  // first maximum wins, like torch.argmax
      bv = acc[c];
      best = c;
    }
  out[(long)y * W0 + x] = (uint8_t)best;
}

// torch's nearest (legacy "nearest", not "nearest-exact"): src = min(floor(dst * in/out), in-1)
__global__ void label_resize_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, uint8_t* __restrict__ dst, int Hd,
                                    int Wd, float sh, float sw, int flip) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= Wd || y >= Hd) return;
  int sy = (int)floorf((float)y * sh), sx = (int)floorf((float)x * sw);
  sy = sy < Hs - 1 ? sy : Hs - 1;
  sx = sx < Ws - 1 ? sx : Ws - 1;
  if (flip) sx = Ws - 1 - sx;
  dst[(long)y * Wd + x] = src[(long)sy * Ws + sx];
}

}  // namespace

extern "C" int rmem_labels_from_logits(const rmem_label_src* srcs, int32_t n_src, int32_t C, int32_t align_corners,
                                       int32_t H0, int32_t W0, uint8_t* label, void* stream) {
  if (!srcs || !label || n_src < 1 || n_src > MAXSRC || C < 1 || C > MAXC || H0 < 1 || W0 < 1) return RMEM_ERR_INVALID;
  LabelSrcs S{};
  for (int i = 0; i < n_src; ++i) {
    if (!srcs[i].logits || srcs[i].h < 1 || srcs[i].w < 1) return RMEM_ERR_INVALID;
    S.logits[i] = srcs[i].logits;
    S.h[i] = srcs[i].h;
    S.w[i] = srcs[i].w;
    S.flip[i] = srcs[i].flip;
    // torch's area_pixel_compute_scale<float>
    if (align_corners) {
      S.rh[i] = H0 > 1 ? (float)(srcs[i].h - 1) / (float)(H0 - 1) : 0.f;
      S.rw[i] = W0 > 1 ? (float)(srcs[i].w - 1) / (float)(W0 - 1) : 0.f;
    } else {
      S.rh[i] = (float)srcs[i].h / (float)H0;
      S.rw[i] = (float)srcs[i].w / (float)W0;
    }
  }
  dim3 grid((W0 + 63) / 64, (H0 + 3) / 4);
  if (n_src == 1)
    hipLaunchKernelGGL(labels_kernel<1>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), S, n_src, C,
                       align_corners, H0, W0, label);
  else
    hipLaunchKernelGGL(labels_kernel<0>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), S, n_src, C,
                       align_corners, H0, W0, label);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_label_resize_nearest(const uint8_t* src, int32_t Hs, int32_t Ws, uint8_t* dst, int32_t Hd,
                                         int32_t Wd, int32_t flip, void* stream) {
  if (!src || !dst || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1) return RMEM_ERR_INVALID;
  dim3 grid((Wd + 63) / 64, (Hd + 3) / 4);
  hipLaunchKernelGGL(label_resize_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, Hs, Ws, dst, Hd,
                     Wd, (float)Hs / (float)Hd, (float)Ws / (float)Wd, flip);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}
