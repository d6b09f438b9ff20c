// attn_common.h -- helpers shared by the memory-read kernels (attn.hip, fused.hip).
#pragma once
#include "rmem_common.h"

// Key-tile band [t_lo, t_hi) (units of 128 keys) visible under the 15x15 window to query tile
// `qtile` (128 queries): image rows y(q_lo)-7 .. y(q_hi)+7.  Scores, P.V and combine all use this
// function with the same 128-query tile, so what one writes (zeros where masked) is what the
// others read.  At 31x54 tokens a band is 7-8 key tiles (the earlier 256-query granularity, kept
// from a removed 256-row P.V kernel, made it 9-10).
__host__ __device__ inline void band_tiles(int qtile, int N, int h, int w, int& t_lo, int& t_hi) {
  const int q_lo = qtile * 128;
  int q_hi = q_lo + 127;
  if (q_hi > N - 1) q_hi = N - 1;
  int y_lo = q_lo / w - 7;
  if (y_lo < 0) y_lo = 0;
  int y_hi = q_hi / w + 7;
  if (y_hi > h - 1) y_hi = h - 1;
  t_lo = (y_lo * w) / 128;
  t_hi = ((y_hi + 1) * w + 127) / 128;
}

// Logical -> physical slot map held in two 64-bit scalars (8 bits per slot, T <= 16).  A
// per-k-tile `slot_map[t]` read is a dependent global load followed by s_waitcnt vmcnt(0): it
// delays the issue of the tile's staging loads by a full memory latency and drains every load in
// flight, which also defeats any deeper prefetch (measured with tools/ubench/pv_trace).
struct SlotLut {
  unsigned long long w0, w1;
  __device__ __forceinline__ void load(const int* slot_map, int T) {
    w0 = w1 = 0;
    if (!slot_map) {
      w0 = 0x0706050403020100ull;
      w1 = 0x0f0e0d0c0b0a0908ull;
      return;
    }
    for (int t = 0; t < T && t < 16; ++t) {
      const unsigned long long v = (unsigned long long)(slot_map[t] & 0xff);
      if (t < 8) w0 |= v << (8 * t); else w1 |= v << (8 * (t - 8));
    }
    w0 = __builtin_amdgcn_readfirstlane((unsigned)w0) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(w0 >> 32)) << 32);
    w1 = __builtin_amdgcn_readfirstlane((unsigned)w1) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(w1 >> 32)) << 32);
  }
  __device__ __forceinline__ int operator()(int t) const {
    return (int)(((t < 8 ? w0 : w1) >> (8 * (t & 7))) & 0xff);
  }
};

