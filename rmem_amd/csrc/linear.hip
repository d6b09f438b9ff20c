// linear.hip -- rmem_linear: D = act(X . Y^T + bias) with fused multi-destination
// epilogue (fp32 / split-fp16 planes / residual accumulate).  See include/rmem_hip.h.
#include "../../include/rmem_hip.h"
#include "gemm_core.h"
#include "launch.h"
#include <stdlib.h>
#include <string.h>

// Epilogue of one output tile of problem `a` (bias, activation, the fp32 / plane / blocked-16 / split-K destinations).
// Cfg gives the wave layout: wave (wr, wc) owns TM x TN accumulator tiles of 32 x 32 at rows wr * WM + tm * 32, columns
// wc * WN + tn * 32 of the block tile at (m0, n0).  Shared by the 4-wave kernels (GemmCfg) and the 8-wave streaming
// kernel (StreamCfg): the same instructions per element whatever kernel computed the accumulators.
template <class Cfg>
__device__ __forceinline__ void linear_epilogue(const rmem_linear_args& a, f32x16_t (&acc)[Cfg::TM][Cfg::TN], int m0, int n0,
                                                int bzz, int wr, int wc, int lane) {
  const bool splitk = a.ksplits > 1;
  const int bz = splitk ? 0 : bzz;
  if (splitk) {   // raw partials [split][M][N]; the consumer (rmem_layernorm_red) sums them in order
    float* out = a.parts + (long)bzz * a.part_stride;
    const float* b0 = (a.bias && bzz == 0) ? a.bias : nullptr;
#pragma unroll
    for (int tn = 0; tn < Cfg::TN; ++tn) {
      const int col = n0 + frag_col<Cfg>(wc, tn, lane);
      if (col >= a.N) continue;
      const float bc = b0 ? b0[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + frag_row<Cfg>(wr, tm, r, lane);
          if (row < a.M) out[(long)row * a.N + col] = acc[tm][tn][r] + bc;
        }
    }
    return;
  }
  const float* bias = a.bias ? a.bias + bz * a.bsbias : nullptr;
  float* d0 = a.d0 ? a.d0 + bz * a.bsd : nullptr;
  float* d1 = a.d1 ? a.d1 + bz * a.bsd : nullptr;
  h16_t* pah = a.pah ? a.pah + bz * a.bspa : nullptr;
  h16_t* pal = a.pal ? a.pal + bz * a.bspa : nullptr;
  h16_t* pbh = a.pbh ? a.pbh + bz * a.bspa : nullptr;
  h16_t* pbl = a.pbl ? a.pbl + bz * a.bspa : nullptr;

#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = n0 + frag_col<Cfg>(wc, tn, lane);
    if (col >= a.N) continue;
    const float bcol = (bias && !a.bias_per_row) ? bias[col] : 0.f;
    const float addv = a.addvec ? a.addvec[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm) {
      if (a.pa_blocked) {
        // plane output in the "blocked-16" layout of the fused memory read (read64.hip): element
        // (row, col) at ((row / 16) * ldpa + col) * 16 + row % 16.  Accumulator registers 4g..4g+3
        // are 4 consecutive rows of one 16-row block: one 8-byte store per plane.
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row0 = m0 + frag_row<Cfg>(wr, tm, 4 * g, lane);
          h16_t hh[4], ll[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[tm][tn][4 * g + e] + bcol;
            if (bias && a.bias_per_row && row0 + e < a.M) v += bias[row0 + e];
            if (a.act == 1) v = silu_f(v);
            split_f16(v, hh[e], ll[e]);
          }
          const long off = ((long)(row0 >> 4) * a.ldpa + col) * 16 + (row0 & 15);
          if (row0 + 3 < a.M) {
            uint2 wh, wl;
            wh.x = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16);
            wh.y = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
            wl.x = (uint32_t)ll[0] | ((uint32_t)ll[1] << 16);
            wl.y = (uint32_t)ll[2] | ((uint32_t)ll[3] << 16);
            *reinterpret_cast<uint2*>(pah + off) = wh;
            if (pal) *reinterpret_cast<uint2*>(pal + off) = wl;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (row0 + e < a.M) {
                pah[off + e] = hh[e];
                if (pal) pal[off + e] = ll[e];
              }
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + frag_row<Cfg>(wr, tm, r, lane);
        if (row >= a.M) continue;
        float v = acc[tm][tn][r] + bcol;
        if (bias && a.bias_per_row) v += bias[row];
        if (a.act == 1) v = silu_f(v);
        if (col < a.csplit) {
          if (d0) {
            float* p = d0 + (long)row * a.ldd0 + (long)col * (a.d0_cs > 0 ? a.d0_cs : 1);
            *p = a.accumulate ? (*p + v) : v;
          }
        } else if (d1) {
          float* p = d1 + (long)row * a.ldd1 + (col - a.csplit);
          *p = a.accumulate ? (*p + v) : v;
        }
        if (pah) {
          h16_t hi, lo;
          split_f16(v, hi, lo);
          pah[(long)row * a.ldpa + col] = hi;
          if (pal) pal[(long)row * a.ldpa + col] = lo;
        }
        if (pbh) {
          h16_t hi, lo;
          split_f16(v + addv, hi, lo);
          pbh[(long)row * a.ldpb + col] = hi;
          if (pbl) pbl[(long)row * a.ldpb + col] = lo;
        }
      }
    }
  }
}

// One BM x BN output tile of problem `a`.  bz = batch index (nbatch > 1) or K-split index
// (ksplits > 1, raw partial sums to a.parts, bias added by split 0 only).
template <int BM, int BN, int NS>
__device__ __forceinline__ void linear_body(const rmem_linear_args& a, int mx, int ny, int bzz, char* smem) {
  using Cfg = GemmCfg<BM, BN, NS>;
  const bool splitk = a.ksplits > 1;
  const int bz = splitk ? 0 : bzz;
  const int m0 = mx * BM, n0 = ny * BN;

  RowMajorOperand lx, ly;
  lx.hi = a.xh + bz * a.bsx;
  lx.lo = a.xl ? a.xl + bz * a.bsx : nullptr;
  lx.ld = a.ldx;
  lx.hi2 = a.xh2 ? a.xh2 + bz * a.bsx : nullptr;
  lx.lo2 = a.xl2 ? a.xl2 + bz * a.bsx : nullptr;
  lx.ld2 = a.ldx2;
  lx.kt_split = a.xh2 ? a.kx_split / 64 : (1 << 30);
  lx.row0 = m0;
  lx.rows = a.M;
  ly.hi = a.yh + bz * a.bsy;
  ly.lo = a.yl ? a.yl + bz * a.bsy : nullptr;
  ly.ld = a.ldy;
  ly.hi2 = a.yh2 ? a.yh2 + bz * a.bsy : nullptr;
  ly.lo2 = a.yl2 ? a.yl2 + bz * a.bsy : nullptr;
  ly.ld2 = a.ldy2;
  ly.kt_split = a.yh2 ? a.ky_split / 64 : (1 << 30);
  ly.row0 = n0;
  ly.rows = a.N;

  GemmFrag<Cfg> f;
  f.zero();
  int kt0 = 0, kt1 = a.K / 64;
  if (splitk) {
    const int per = (kt1 + a.ksplits - 1) / a.ksplits;
    kt0 = bzz * per;
    kt1 = kt0 + per < kt1 ? kt0 + per : kt1;
  }
  gemm_mainloop<Cfg>(f, lx, ly, kt0, kt1, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  linear_epilogue<Cfg>(a, f.acc, m0, n0, bzz, wave >> 1, wave & 1, lane);
}

template <int BM, int BN, int NS>
__device__ void linear_kernel(const rmem_linear_args& a, int bz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  linear_body<BM, BN, NS>(a, blockIdx.x, blockIdx.y, bz, smem);
}

// Several independent small problems in one launch (the projections of one LSTT stage share
// their input): 16-byte kernel boundaries and 54-216-block grids are replaced by one grid
// that fills the chip.  tile_start[i] = first block of problem i.
struct GroupedLinear {
  int n;
  int tile_start[9];
  rmem_linear_args p[8];
};

template <int NS, bool DEV>
__device__ void linear_grouped_body(const GroupedLinear& g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.tile_start[i + 1]) ++i;
  // DEV: the group lives in device memory (several clips per launch, launch.h): the selected problem
  // is copied through scalar loads so that its fields sit in SGPRs like kernel arguments
  rmem_linear_args acopy;
  if constexpr (DEV) acopy = rmem::uniform_copy(&g.p[__builtin_amdgcn_readfirstlane(i)]);
  const rmem_linear_args& a = DEV ? acopy : g.p[i];
  int local = blockIdx.x - g.tile_start[i];
  const int mt = (a.M + 63) / 64, nt = (a.N + 63) / 64;
  const int bz = local / (mt * nt);
  local -= bz * mt * nt;
  const int ny = local / mt, mx = local - ny * mt;
  linear_body<64, 64, NS>(a, mx, ny, bz, smem);
}
template <int NS>
__device__ void linear_grouped_kernel(const GroupedLinear& g, int) { linear_grouped_body<NS, false>(g); }
template <int NS>
__device__ void linear_grouped_many(const GroupedLinear& g, int) { linear_grouped_body<NS, true>(g); }

static int validate_linear(rmem_linear_args& a);

#include "linear_stream.h"

// Debug aid: the streaming kernel with cycle stamps (see linear_stream_kernel); trace must hold 64 int64 per workgroup
// (at most one per CU).  nsplit = 3 only.
extern "C" int rmem_linear_trace(const rmem_linear_args* args, int32_t n, int64_t* trace, void* stream) {
  if (!args || n <= 0 || n > 8 || !trace) return RMEM_ERR_INVALID;
  rmem_linear_args v[8];
  for (int i = 0; i < n; ++i) {
    v[i] = args[i];
    if (validate_linear(v[i]) != RMEM_OK || v[i].nsplit != 3) return RMEM_ERR_INVALID;
  }
  if (!use_stream(v, n)) return RMEM_ERR_INVALID;
  StreamGroup2 g2;
  const int total2 = stream2_group(v, n, g2);
  if (!stream2_covers(g2)) return RMEM_ERR_INVALID;
  const int var = rmem_config().stream_var;          // timing experiments (linear_stream2_kernel): 2 no requests, 3 no MFMAs, 4 no fragment reads
  long long* tp = reinterpret_cast<long long*>(trace);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (var == 2) return launch_stream2<3, 2>(g2, total2, tp, st);
  if (var == 3) return launch_stream2<3, 3>(g2, total2, tp, st);
  if (var == 4) return launch_stream2<3, 4>(g2, total2, tp, st);
  return launch_stream2<3, 1>(g2, total2, tp, st);
}

template <int BM, int BN, int NS>
static int launch_linear(const rmem_linear_args& a, hipStream_t s) {
  using Cfg = GemmCfg<BM, BN, NS>;
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.ksplits > 1 ? a.ksplits : (a.nbatch > 0 ? a.nbatch : 1));
  return rmem::launch<rmem_linear_args, linear_kernel<BM, BN, NS>, 256>(a, grid, dim3(256), Cfg::LDS_BYTES, s);
}

static int validate_linear(rmem_linear_args& a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % 64) != 0) return RMEM_ERR_INVALID;
  if (!a.xh || !a.yh) return RMEM_ERR_INVALID;
  if (a.nsplit != 1 && a.nsplit != 3) return RMEM_ERR_INVALID;
  if (a.nsplit == 3 && (!a.xl || !a.yl)) return RMEM_ERR_INVALID;
  if (a.xh2 && ((a.kx_split % 64) != 0 || a.kx_split <= 0 || a.kx_split >= a.K)) return RMEM_ERR_INVALID;
  if (a.yh2 && ((a.ky_split % 64) != 0 || a.ky_split <= 0 || a.ky_split >= a.K)) return RMEM_ERR_INVALID;
  if ((a.ldx % 8) || (a.ldy % 8)) return RMEM_ERR_INVALID;
  if (a.csplit <= 0 || a.csplit > a.N) a.csplit = a.N;
  if (a.ksplits > 1 && (!a.parts || a.nbatch > 1 || a.act != 0 || a.bias_per_row || a.ksplits > a.K / 64))
    return RMEM_ERR_INVALID;
  if (a.pa_blocked && (!a.pah || a.d0 || a.d1 || a.pbh || a.ksplits > 1)) return RMEM_ERR_INVALID;   // blocked planes are the only output
  return RMEM_OK;
}

extern "C" int rmem_linear_grouped(const rmem_linear_args* args, int32_t n, void* stream) {
  if (!args || n <= 0 || n > 8) return RMEM_ERR_INVALID;
  GroupedLinear g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = args[i];
    if (validate_linear(g.p[i]) != RMEM_OK || g.p[i].nsplit != args[0].nsplit) return RMEM_ERR_INVALID;
    const rmem_linear_args& a = g.p[i];
    g.tile_start[i] = total;
    total += ((a.M + 63) / 64) * ((a.N + 63) / 64) * (a.ksplits > 1 ? a.ksplits : (a.nbatch > 0 ? a.nbatch : 1));
  }
  g.tile_start[n] = total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (use_stream(g.p, n)) {
    StreamGroup2 g2;
    const int tot2 = stream2_group(g.p, n, g2);
    if (stream2_covers(g2))
      return args[0].nsplit == 3 ? launch_stream2<3, 0>(g2, tot2, nullptr, s) : launch_stream2<1, 0>(g2, tot2, nullptr, s);
    // (a problem with a generic epilogue shape -- per-row bias, accumulate, two fp32 destinations: the tile kernels below)
  }
  if (args[0].nsplit == 3)
    return rmem::launch<GroupedLinear, linear_grouped_kernel<3>, 256, linear_grouped_many<3>>(
        g, dim3(total), dim3(256), GemmCfg<64, 64, 3>::LDS_BYTES, s);
  return rmem::launch<GroupedLinear, linear_grouped_kernel<1>, 256, linear_grouped_many<1>>(
      g, dim3(total), dim3(256), GemmCfg<64, 64, 1>::LDS_BYTES, s);
}

extern "C" int rmem_linear(const rmem_linear_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  rmem_linear_args a = *ap;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (validate_linear(a) != RMEM_OK) return RMEM_ERR_INVALID;
  if (use_stream(&a, 1)) {
    StreamGroup2 g2;
    const int tot2 = stream2_group(&a, 1, g2);
    if (stream2_covers(g2)) return a.nsplit == 3 ? launch_stream2<3, 0>(g2, tot2, nullptr, s) : launch_stream2<1, 0>(g2, tot2, nullptr, s);
  }
  // tile-per-workgroup kernels: 64 x 64 unless 64 rows x 128 columns is asked for (the 128 x 128 instantiation -- 248
  // registers + 64 accumulator registers, 336 B of scratch, one wave per SIMD, never selected by the memory path -- is gone)
  const int tile = (a.tile == 0 || a.tile == 256 || a.tile == 128) ? 64 : a.tile;
  if (tile == 64) return a.nsplit == 3 ? launch_linear<64, 64, 3>(a, s) : launch_linear<64, 64, 1>(a, s);
  if (tile == 192) return a.nsplit == 3 ? launch_linear<64, 128, 3>(a, s) : launch_linear<64, 128, 1>(a, s);   // 64 rows x 128 columns
  return RMEM_ERR_INVALID;
}
