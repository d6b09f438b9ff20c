// attn_common.h -- helpers of the memory-read kernel (read64.hip).
#pragma once
#include "rmem_common.h"

// Logical -> physical slot map held in two 64-bit scalars (8 bits per slot, T <= 16).  A
// per-k-tile `slot_map[t]` read is a dependent global load followed by s_waitcnt vmcnt(0): it
// delays the issue of the tile's staging loads by a full memory latency and drains every load in
// flight, which also defeats any deeper prefetch (measured with a tracing copy of the round-2 kernel: git history, research/ubench/pv_trace.hip).
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

