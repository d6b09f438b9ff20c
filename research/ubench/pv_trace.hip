// pv_trace.hip -- standalone micro-benchmark of the long-term P.V kernel (480p K=4 problem) with
// per-k-step shader-clock stamps of wave 0 of a few blocks.  Shows where a k-step's cycles go
// (barrier waits, global-load landing, LDS staging, fragment reads + MFMA).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 research/ubench/pv_trace.hip -o research/ubench/pv_trace
#include "../../rmem_amd/csrc/attn.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define NSTAMP 6

template <class Cfg, class LX, class LY>
__device__ __forceinline__ void gemm_mainloop_traced(GemmFrag<Cfg>& f, const LX& lx, const LY& ly, int kt_begin,
                                                     int kt_end, char* smem, long long* stamps, int rot = 0) {
  constexpr int NPL = Cfg::NPL, XCH = Cfg::XCH, YCH = Cfg::YCH, TM = Cfg::TM, TN = Cfg::TN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (kt_begin >= kt_end) return;
  u32x4_t xr[NPL * XCH], yr[NPL * YCH];
  auto gload = [&](int kt) __attribute__((always_inline)) {
    const TileView tx = lx.tile(kt);
    const TileView ty = ly.tile(kt);
    static_for<NPL>([&](auto P) {
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        xr[P.value * XCH + I.value] = *lx.ptr(tx, P.value, id >> 3, id & 7);
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        yr[P.value * YCH + I.value] = *ly.ptr(ty, P.value, id >> 3, id & 7);
      });
    });
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    static_for<NPL>([&](auto P) {
      char* xb = smem + P.value * Cfg::X_BYTES;
      char* yb = smem + NPL * Cfg::X_BYTES + P.value * Cfg::Y_BYTES;
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(xb + lds_swz(id >> 3, id & 7)) = xr[P.value * XCH + I.value];
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(yb + lds_swz(id >> 3, id & 7)) = yr[P.value * YCH + I.value];
      });
    });
  };
  auto stamp = [&](int kt, int i) __attribute__((always_inline)) {
    if (stamps && tid == 0) stamps[(kt - kt_begin) * NSTAMP + i] = clock64();
  };
  const int nkt = kt_end - kt_begin;
  auto rotk = [&](int kt) { int i = kt - kt_begin + rot; i = i >= nkt ? i - nkt : i; return kt_begin + i; };
  gload(rotk(kt_begin));
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    stamp(kt, 0);
    __syncthreads();
    stamp(kt, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(kt, 2);
    lstore();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(kt, 3);
    __syncthreads();
    stamp(kt, 4);
    if (kt + 1 < kt_end) gload(rotk(kt + 1));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      bf16x8_t a[NPL][TM], b[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const char* xb = smem + p * Cfg::X_BYTES;
        const char* yb = smem + NPL * Cfg::X_BYTES + p * Cfg::Y_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * Cfg::WM + i * 32 + (lane & 31);
          a[p][i] = *reinterpret_cast<const bf16x8_t*>(xb + lds_swz(row, chunk));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wc * Cfg::WN + j * 32 + (lane & 31);
          b[p][j] = *reinterpret_cast<const bf16x8_t*>(yb + lds_swz(row, chunk));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (Cfg::NSPLIT == 3) {
            f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], f.acc[i][j], 0, 0, 0);
            f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], f.acc[i][j], 0, 0, 0);
          }
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], f.acc[i][j], 0, 0, 0);
        }
    }
    stamp(kt, 5);
  }
}

template <int NS>
__global__ __launch_bounds__(256) void pv_kernel_traced(rmem_pv_args a, long long* stamps, int nsteps_max, int delay, int pattern) {
  using Cfg = GemmCfg<128, 128, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 127) / 128;
  const int nq = a.Npad / 128;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pl = j / nct;
  const int ctile = j - pl * nct;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps = a.Npad / 64;
  const int k_lo = 0, k_hi = a.T * tps;
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  int lo = k_lo + z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;
  PBlockedOperand lx{a.ph, a.pl, (long)a.Npad, qtile * 128};
  SlotLut lut;
  lut.load(a.slot_map, a.T);
  VtOperand ly{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols};
  GemmFrag<Cfg> f;
  f.zero();
  // phase offset between the blocks that share a CU (pattern 0: second dispatch round, 1: odd j)
  const bool late = pattern == 0 ? (j >= 32) : (j & 1);
  if (late) for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16);
  long long* st = nullptr;
  if (stamps && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == 300))
    st = stamps + (blockIdx.x == 0 ? 0 : blockIdx.x == 101 ? 1 : 2) * nsteps_max * NSTAMP;
  int rot = 0;
  const int nkt_ = hi - lo;
  if (pattern == 10) rot = (ctile * nkt_) / nct;
  if (pattern == 11) rot = ((ctile * 14 + qtile) * nkt_ / (nct * 14));
  if (pattern == 12) rot = (blockIdx.x * 7) % (nkt_ > 0 ? nkt_ : 1);
  gemm_mainloop_traced<Cfg>(f, lx, ly, lo, hi, smem, st, rot);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float* out = a.part + (long)z * a.Npad * a.ncols;
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = ctile * 128 + frag_col<Cfg>(wc, tn, lane);
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 128 + frag_row<Cfg>(wr, tm, r, lane);
        out[(long)q * a.ncols + col] = f.acc[tm][tn][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// V2: 32-key steps, LDS double buffer (same 64 KB as the shipped loop -> still 2 blocks per CU),
// ONE barrier per step, tile k+2's global loads issued at the top of step k (a full step and a
// half to land), tile k+1's LDS staging placed between the two MFMA sub-steps of tile k.
template <class Cfg, class LX, class LY>
__device__ __forceinline__ void gemm_mainloop_v2(GemmFrag<Cfg>& f, const LX& lx, const LY& ly, int kt_begin,
                                                 int kt_end, char* smem) {
  // kt counts 64-key tiles as in the shipped loop; internally every tile is two 32-key half-steps.
  constexpr int NPL = Cfg::NPL, TM = Cfg::TM, TN = Cfg::TN;
  constexpr int XCH = Cfg::BM * 4 / Cfg::THREADS, YCH = Cfg::BN * 4 / Cfg::THREADS;   // 16-B chunks per thread per plane per half-step
  constexpr int HX = Cfg::BM * 64, HY = Cfg::BN * 64;                                  // bytes per plane per half-step (64-byte rows)
  constexpr int STAGE = NPL * (HX + HY);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (kt_begin >= kt_end) return;
  const int h_begin = 2 * kt_begin, h_end = 2 * kt_end;
  u32x4_t ra[NPL * (XCH + YCH)], rb[NPL * (XCH + YCH)];
  // LDS image of a half-step plane: row r = 64 bytes = 4 chunks; chunk c stored at c ^ ((r >> 2) & 3)
  auto swz = [](int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); };
  auto gload = [&](int hs, u32x4_t* r) __attribute__((always_inline)) {
    const int kt = hs >> 1, half = hs & 1;
    const TileView tx = lx.tile(kt);
    const TileView ty = ly.tile(kt);
    static_for<NPL>([&](auto P) {
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        r[P.value * (XCH + YCH) + I.value] = *lx.ptr(tx, P.value, id >> 2, half * 4 + (id & 3));
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        r[P.value * (XCH + YCH) + XCH + I.value] = *ly.ptr(ty, P.value, id >> 2, half * 4 + (id & 3));
      });
    });
  };
  auto lstore = [&](char* st, const u32x4_t* r) __attribute__((always_inline)) {
    static_for<NPL>([&](auto P) {
      char* xb = st + P.value * HX;
      char* yb = st + NPL * HX + P.value * HY;
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(xb + swz(id >> 2, id & 3)) = r[P.value * (XCH + YCH) + I.value];
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(yb + swz(id >> 2, id & 3)) = r[P.value * (XCH + YCH) + XCH + I.value];
      });
    });
  };
  auto substep = [&](const char* st, int ks) __attribute__((always_inline)) {
    const int chunk = ks * 2 + (lane >> 5);
    bf16x8_t a[NPL][TM], b[NPL][TN];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const char* xb = st + p * HX;
      const char* yb = st + NPL * HX + p * HY;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * Cfg::WM + i * 32 + (lane & 31);
        a[p][i] = *reinterpret_cast<const bf16x8_t*>(xb + swz(row, chunk));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * Cfg::WN + j * 32 + (lane & 31);
        b[p][j] = *reinterpret_cast<const bf16x8_t*>(yb + swz(row, chunk));
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (Cfg::NSPLIT == 3) {
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], f.acc[i][j], 0, 0, 0);
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], f.acc[i][j], 0, 0, 0);
        }
        f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], f.acc[i][j], 0, 0, 0);
      }
  };
  // prologue: half-step h_begin -> LDS stage 0, h_begin+1 in flight in ra.  The number of
  // half-steps is even; loads past the end are clamped to the last half-step (never consumed), so
  // the loop body is branch-free and the compiler counts outstanding loads exactly (with
  // conditional loads it falls back to vmcnt(7..0) and waits for the newest tile).
  const int h_last = h_end - 1;
  auto clampi = [&](int h) { return h < h_last ? h : h_last; };
  gload(h_begin, ra);
  lstore(smem, ra);
  gload(h_begin + 1, ra);
  for (int hs = h_begin; hs < h_end; hs += 2) {
    __syncthreads();                                   // stage 0 holds hs; stage 1 free
    gload(clampi(hs + 2), rb);
    substep(smem, 0);
    lstore(smem + STAGE, ra);                          // hs+1 (loaded one step ago)
    substep(smem, 1);
    __syncthreads();                                   // stage 1 holds hs+1; stage 0 free
    gload(clampi(hs + 3), ra);
    substep(smem + STAGE, 0);
    lstore(smem, rb);
    substep(smem + STAGE, 1);
  }
}

template <int NS>
__global__ __launch_bounds__(256) void pv_kernel_v2(rmem_pv_args a) {
  using Cfg = GemmCfg<128, 128, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 127) / 128;
  const int nq = a.Npad / 128;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pl = j / nct;
  const int ctile = j - pl * nct;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps = a.Npad / 64;
  const int k_lo = 0, k_hi = a.T * tps;
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  int lo = k_lo + z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;
  PBlockedOperand lx{a.ph, a.pl, (long)a.Npad, qtile * 128};
  SlotLut lut;
  lut.load(a.slot_map, a.T);
  VtOperand ly{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols};
  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop_v2<Cfg>(f, lx, ly, lo, hi, smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float* out = a.part + (long)z * a.Npad * a.ncols;
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = ctile * 128 + frag_col<Cfg>(wc, tn, lane);
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 128 + frag_row<Cfg>(wr, tm, r, lane);
        out[(long)q * a.ncols + col] = f.acc[tm][tn][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// V3: 256 (queries) x 128 (columns) tile, 8 waves (4 x 2, wave tile 64 x 64), 32-key steps staged
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, source-side swizzle) into a ring of three
// 48 KB stages, two steps of prefetch, counted vmcnt, one raw barrier per step.  One block per CU.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int NS, int VAR>
__global__ __launch_bounds__(512) void pv_kernel_v3(rmem_pv_args a, long long* stamps) {
  constexpr int NPL = (NS == 1) ? 1 : 2;
  constexpr int PX = 256 * 64, PY = 128 * 64;           // bytes per plane per stage
  constexpr int STAGE = NPL * (PX + PY);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 127) / 128;
  const int nq = a.Npad / 256;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pl = j / nct;
  const int ctile = j - pl * nct;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps32 = a.Npad / 32;
  const int k_hi = a.T * tps32;
  const int per = (k_hi + a.ksplits - 1) / a.ksplits;
  int lo = z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  SlotLut lut;
  lut.load(a.slot_map, a.T);

  // DMA sources of this lane: P groups 2*wave, 2*wave+1 (16 rows each), V group wave
  long xoff[2], yoff;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (2 * wave + u) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    xoff[u] = (long)(qtile * 256 + r) * 32 + c * 8;
  }
  {
    const int r = wave * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    int jrow = ctile * 128 + r;
    jrow = jrow < a.ncols ? jrow : a.ncols - 1;
    yoff = (long)jrow * a.Npad + c * 8;
  }
  const bf16_t* xpl[2] = {a.ph, a.pl};
  const bf16_t* ypl[2] = {a.vh, a.vl};
  auto issue = [&](int kb, char* sb) __attribute__((always_inline)) {
    const int t = kb / tps32;
    const long xbase = (long)kb * a.Npad * 32;
    const long ybase = (long)lut(t) * a.v_slot_stride + (long)(kb - t * tps32) * 32;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(xpl[p] + xbase + xoff[u]),
                                         (lds_void_t*)(sb + p * PX + (2 * wave + u) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(ypl[p] + ybase + yoff),
                                       (lds_void_t*)(sb + NPL * PX + p * PY + wave * 1024), 16, 0, 0);
    }
  };
  int xfo[2][2], yfo[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = ks * 2 + (lane >> 5);
      const int rx = wr * 64 + i * 32 + (lane & 31);
      const int ry = wc * 64 + i * 32 + (lane & 31);
      xfo[ks][i] = rx * 64 + ((c ^ ((rx >> 2) & 3)) << 4);
      yfo[ks][i] = ry * 64 + ((c ^ ((ry >> 2) & 3)) << 4);
    }
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  bf16x8_t xa[2][NPL][2], yb[2][NPL][2];
  auto readfrags = [&](const char* sb) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          xa[ks][p][i] = *reinterpret_cast<const bf16x8_t*>(sb + p * PX + xfo[ks][i]);
          yb[ks][p][i] = *reinterpret_cast<const bf16x8_t*>(sb + NPL * PX + p * PY + yfo[ks][i]);
        }
  };
  auto mfmas = [&]() __attribute__((always_inline)) {
    if constexpr (VAR >= 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if constexpr (NS == 3) {
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks][0][i], yb[ks][1][jj], acc[i][jj], 0, 0, 0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks][1][i], yb[ks][0][jj], acc[i][jj], 0, 0, 0);
          }
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[ks][0][i], yb[ks][0][jj], acc[i][jj], 0, 0, 0);
        }
    if constexpr (VAR >= 2) __builtin_amdgcn_s_setprio(0);
  };
  const int last = hi - 1;
  auto clampi = [&](int k) { return k < last ? k : last; };
  constexpr int PER = 3 * NPL;                           // DMA instructions per wave per stage
  // prologue: steps lo, lo+1 in flight
  issue(clampi(lo), smem);
  issue(clampi(lo + 1), smem + STAGE);
  long long* st = (stamps && blockIdx.x == 8 && (tid & 63) == 0) ? stamps + (wave * 128) * 4 : nullptr;
  auto step = [&](int kb, char* cur, char* nxt2) __attribute__((always_inline)) {
    // the batch of step kb is the older of the two in flight
    if (st) st[(kb - lo) * 4 + 0] = clock64();
    if constexpr (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (st) st[(kb - lo) * 4 + 1] = clock64();
    __builtin_amdgcn_s_barrier();                        // kb landed everywhere; stage of kb-1 is free
    if (st) st[(kb - lo) * 4 + 2] = clock64();
    if constexpr (VAR == 0) {
      issue(clampi(kb + 2), nxt2);
      readfrags(cur);
    } else {
      readfrags(cur);
      __builtin_amdgcn_sched_barrier(0);
      issue(clampi(kb + 2), nxt2);
    }
    mfmas();
    if (st) st[(kb - lo) * 4 + 3] = clock64();
  };
  int kb = lo;
  for (; kb + 2 < hi; kb += 3) {
    step(kb, smem, smem + 2 * STAGE);
    step(kb + 1, smem + STAGE, smem);
    step(kb + 2, smem + 2 * STAGE, smem + STAGE);
  }
  if (kb < hi) { step(kb, smem, smem + 2 * STAGE); ++kb; }
  if (kb < hi) { step(kb, smem + STAGE, smem); ++kb; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* out = a.part + (long)z * a.Npad * a.ncols;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int col = ctile * 128 + wc * 64 + tn * 32 + (lane & 31);
    if (col >= a.ncols) continue;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 256 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(long)q * a.ncols + col] = acc[tm][tn][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// V5: 256 x 256 tile, 8 waves (2 x 4, wave tile 128 x 64), 16-key steps (= one MFMA k-step) staged
// by LDS-DMA into a ring of four 32 KB stages (three steps of prefetch, counted vmcnt).
// STAGGER: the two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run half a step
// apart (group 1 executes one extra barrier up front): a step is a read phase R (wait for the own
// DMA part, fragment reads, DMA issue) and an MFMA phase M, each closed by a barrier, so that on
// every SIMD one wave is in M while the other is in R.
template <int NS, bool STAGGER>
__global__ __launch_bounds__(512) void pv_kernel_v5(rmem_pv_args a, long long* stamps) {
  static_assert(NS == 3, "planes");
  constexpr int NPL = 2;
  constexpr int PX = 256 * 32, PY = 256 * 32;           // bytes per plane per stage (32-byte rows)
  constexpr int STAGE = NPL * (PX + PY);                // 32 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 255) / 256;
  const int nq = (a.Npad + 255) / 256;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
  const int pl = j0 / nct;
  const int ctile = j0 - pl * nct;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps16 = a.Npad / 16;
  const int k_hi = a.T * tps16;
  int per = (k_hi + a.ksplits - 1) / a.ksplits;
  per = (per + 3) & ~3;                                 // whole 64-key tiles per split, like pv_kernel
  int lo = z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  SlotLut lut;
  lut.load(a.slot_map, a.T);
  float* out = a.part + (long)z * a.Npad * a.ncols;
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

  if (lo < hi) {
    // DMA sources of this lane: rows [32 wave, 32 wave + 32) of each of the four plane tiles
    const int srow = 32 * wave + (lane >> 1);
    const int sc = (lane & 1) ^ ((srow >> 3) & 1);
    long q = qtile * 256 + srow;
    q = q < a.Npad ? q : a.Npad - 1;
    const long xoff = q * 32 + sc * 8;
    int jrow = ctile * 256 + srow;
    jrow = jrow < a.ncols ? jrow : a.ncols - 1;
    const long yoff = (long)jrow * a.Npad + sc * 8;
    auto issue = [&](int kb, char* sb) __attribute__((always_inline)) {
      const int t = kb / tps16;
      const long xbase = (long)(kb >> 1) * a.Npad * 32 + (kb & 1) * 16;
      const long ybase = (long)lut(t) * a.v_slot_stride + (long)(kb - t * tps16) * 16;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.ph + xbase + xoff), (lds_void_t*)(sb + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.pl + xbase + xoff), (lds_void_t*)(sb + PX + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.vh + ybase + yoff), (lds_void_t*)(sb + 2 * PX + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.vl + ybase + yoff), (lds_void_t*)(sb + 2 * PX + PY + wave * 1024), 16, 0, 0);
    };
    int xfo[4], yfo[2];
    {
      const int c = lane >> 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rx = wr * 128 + i * 32 + (lane & 31);
        xfo[i] = rx * 32 + ((c ^ ((rx >> 3) & 1)) << 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ry = wc * 64 + i * 32 + (lane & 31);
        yfo[i] = ry * 32 + ((c ^ ((ry >> 3) & 1)) << 4);
      }
    }
    bf16x8_t xa[NPL][4], yb[NPL][2];
    auto readfrags = [&](const char* sb) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) yb[p][i] = *reinterpret_cast<const bf16x8_t*>(sb + 2 * PX + p * PY + yfo[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[p][i] = *reinterpret_cast<const bf16x8_t*>(sb + p * PX + xfo[i]);
      }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][i], yb[1][jj], acc[i][jj], 0, 0, 0);
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1][i], yb[0][jj], acc[i][jj], 0, 0, 0);
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][i], yb[0][jj], acc[i][jj], 0, 0, 0);
        }
    };
    const int last = hi - 1;
    auto clampi = [&](int k) { return k < last ? k : last; };
    long long* st = (stamps && blockIdx.x == 8 && (tid & 63) == 0) ? stamps + (wave * 128) * 4 : nullptr;
    issue(clampi(lo), smem);
    issue(clampi(lo + 1), smem + STAGE);
    issue(clampi(lo + 2), smem + 2 * STAGE);
    if constexpr (STAGGER) {
      if (wr == 1) __builtin_amdgcn_s_barrier();
    }
    auto step = [&](int kb, char* cur, char* nxt3) __attribute__((always_inline)) {
      if (st && kb - lo < 128) st[(kb - lo) * 4 + 0] = clock64();
      if constexpr (STAGGER) {
        // R: own part of batch kb+1 landed (kb itself was confirmed one step ago); kb+2 stays in flight
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (st && kb - lo < 128) st[(kb - lo) * 4 + 1] = clock64();
        readfrags(cur);
        issue(clampi(kb + 3), nxt3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st && kb - lo < 128) st[(kb - lo) * 4 + 2] = clock64();
        __builtin_amdgcn_s_setprio(1);
        mfmas();
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
      } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // batch kb landed (kb+1, kb+2 in flight)
        if (st && kb - lo < 128) st[(kb - lo) * 4 + 1] = clock64();
        __builtin_amdgcn_s_barrier();
        if (st && kb - lo < 128) st[(kb - lo) * 4 + 2] = clock64();
        issue(clampi(kb + 3), nxt3);
        readfrags(cur);
        mfmas();
      }
      if (st && kb - lo < 128) st[(kb - lo) * 4 + 3] = clock64();
    };
    if constexpr (STAGGER) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // batch lo landed (own part) ...
    if constexpr (STAGGER) __builtin_amdgcn_s_barrier();                        // ... for everyone (both groups pass one barrier)
    int kb = lo;
    for (; kb + 3 < hi; kb += 4) {
      step(kb, smem, smem + 3 * STAGE);
      step(kb + 1, smem + STAGE, smem);
      step(kb + 2, smem + 2 * STAGE, smem + STAGE);
      step(kb + 3, smem + 3 * STAGE, smem + 2 * STAGE);
    }
    if (kb < hi) { step(kb, smem, smem + 3 * STAGE); ++kb; }
    if (kb < hi) { step(kb, smem + STAGE, smem); ++kb; }
    if (kb < hi) { step(kb, smem + 2 * STAGE, smem + STAGE); ++kb; }
    if constexpr (STAGGER) {
      if (wr == 0) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int col = ctile * 256 + wc * 64 + tn * 32 + (lane & 31);
    if (col >= a.ncols) continue;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 256 + wr * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (q < a.Npad) out[(long)q * a.ncols + col] = acc[tm][tn][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// V6: the shipped 128 x 128 / 4-wave loop, two of them per workgroup (512 threads = two groups,
// each with its own output tile and its own 64 KB of LDS) held in ANTI-PHASE by the workgroup
// barrier: group 1 runs one barrier behind, so one group multiplies while the other stages.
// (Two independent 256-thread blocks per CU lock into phase: both stage, then both multiply --
// measured with the step stamps above -- and the MFMA pipe idles a third of the time.)
template <class Cfg, class LX, class LY>
__device__ __forceinline__ void gemm_mainloop_grp(GemmFrag<Cfg>& f, const LX& lx, const LY& ly, int kt_begin,
                                                  int kt_end, char* smem, int tid, int group) {
  constexpr int NPL = Cfg::NPL, XCH = Cfg::XCH, YCH = Cfg::YCH, TM = Cfg::TM, TN = Cfg::TN;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  u32x4_t xr[NPL * XCH], yr[NPL * YCH];
  auto gload = [&](int kt) __attribute__((always_inline)) {
    const TileView tx = lx.tile(kt);
    const TileView ty = ly.tile(kt);
    static_for<NPL>([&](auto P) {
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        xr[P.value * XCH + I.value] = *lx.ptr(tx, P.value, id >> 3, id & 7);
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        yr[P.value * YCH + I.value] = *ly.ptr(ty, P.value, id >> 3, id & 7);
      });
    });
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    static_for<NPL>([&](auto P) {
      char* xb = smem + P.value * Cfg::X_BYTES;
      char* yb = smem + NPL * Cfg::X_BYTES + P.value * Cfg::Y_BYTES;
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(xb + lds_swz(id >> 3, id & 7)) = xr[P.value * XCH + I.value];
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(yb + lds_swz(id >> 3, id & 7)) = yr[P.value * YCH + I.value];
      });
    });
  };
  if (kt_begin < kt_end) gload(kt_begin);
  if (group == 1) __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    lstore();                                   // stage phase (the other group multiplies)
    __syncthreads();
    if (kt + 1 < kt_end) gload(kt + 1);         // multiply phase (the other group stages)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      bf16x8_t a[NPL][TM], b[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const char* xb = smem + p * Cfg::X_BYTES;
        const char* yb = smem + NPL * Cfg::X_BYTES + p * Cfg::Y_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * Cfg::WM + i * 32 + (lane & 31);
          a[p][i] = *reinterpret_cast<const bf16x8_t*>(xb + lds_swz(row, chunk));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wc * Cfg::WN + j * 32 + (lane & 31);
          b[p][j] = *reinterpret_cast<const bf16x8_t*>(yb + lds_swz(row, chunk));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (Cfg::NSPLIT == 3) {
            f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], f.acc[i][j], 0, 0, 0);
            f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], f.acc[i][j], 0, 0, 0);
          }
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], f.acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  if (group == 0) __syncthreads();
}

template <int NS>
__global__ __launch_bounds__(512) void pv_kernel_v6(rmem_pv_args a) {
  using Cfg = GemmCfg<128, 128, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int group = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int nct = (a.ncols + 127) / 128;       // even
  const int nq = a.Npad / 128;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int nct2 = nct / 2;
  const int pl = j / nct2;
  const int ctile = 2 * (j - pl * nct2) + group;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps = a.Npad / 64;
  const int k_lo = 0, k_hi = a.T * tps;
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  int lo = k_lo + z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;
  PBlockedOperand lx{a.ph, a.pl, (long)a.Npad, qtile * 128};
  SlotLut lut;
  lut.load(a.slot_map, a.T);
  VtOperand ly{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols};
  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop_grp<Cfg>(f, lx, ly, lo, hi, smem + group * Cfg::LDS_BYTES, tid, group);
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float* out = a.part + (long)z * a.Npad * a.ncols;
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = ctile * 128 + frag_col<Cfg>(wc, tn, lane);
    if (col >= a.ncols) continue;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 128 + frag_row<Cfg>(wr, tm, r, lane);
        out[(long)q * a.ncols + col] = f.acc[tm][tn][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// V7: the shipped loop with the fragment reads of sub-step s+1 issued BEFORE the MFMAs of
// sub-step s (two fragment register sets).  PIN: sched_barrier between read group and MFMA group.
template <class Cfg, bool PIN, class LX, class LY>
__device__ __forceinline__ void gemm_mainloop_v7(GemmFrag<Cfg>& f, const LX& lx, const LY& ly, int kt_begin,
                                                 int kt_end, char* smem) {
  constexpr int NPL = Cfg::NPL, XCH = Cfg::XCH, YCH = Cfg::YCH, TM = Cfg::TM, TN = Cfg::TN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (kt_begin >= kt_end) return;
  u32x4_t xr[NPL * XCH], yr[NPL * YCH];
  auto gload = [&](int kt) __attribute__((always_inline)) {
    const TileView tx = lx.tile(kt);
    const TileView ty = ly.tile(kt);
    static_for<NPL>([&](auto P) {
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        xr[P.value * XCH + I.value] = *lx.ptr(tx, P.value, id >> 3, id & 7);
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        yr[P.value * YCH + I.value] = *ly.ptr(ty, P.value, id >> 3, id & 7);
      });
    });
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    static_for<NPL>([&](auto P) {
      char* xb = smem + P.value * Cfg::X_BYTES;
      char* yb = smem + NPL * Cfg::X_BYTES + P.value * Cfg::Y_BYTES;
      static_for<XCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(xb + lds_swz(id >> 3, id & 7)) = xr[P.value * XCH + I.value];
      });
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(yb + lds_swz(id >> 3, id & 7)) = yr[P.value * YCH + I.value];
      });
    });
  };
  bf16x8_t fa[2][NPL][TM], fb[2][NPL][TN];
  auto rd = [&](int ks, auto S) __attribute__((always_inline)) {
    const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const char* xb = smem + p * Cfg::X_BYTES;
      const char* yb = smem + NPL * Cfg::X_BYTES + p * Cfg::Y_BYTES;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * Cfg::WM + i * 32 + (lane & 31);
        fa[S.value][p][i] = *reinterpret_cast<const bf16x8_t*>(xb + lds_swz(row, chunk));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * Cfg::WN + j * 32 + (lane & 31);
        fb[S.value][p][j] = *reinterpret_cast<const bf16x8_t*>(yb + lds_swz(row, chunk));
      }
    }
  };
  auto mm = [&](auto S) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (Cfg::NSPLIT == 3) {
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S.value][0][i], fb[S.value][1][j], f.acc[i][j], 0, 0, 0);
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S.value][1][i], fb[S.value][0][j], f.acc[i][j], 0, 0, 0);
        }
        f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S.value][0][i], fb[S.value][0][j], f.acc[i][j], 0, 0, 0);
      }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  gload(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    rd(0, S0{});
    if (kt + 1 < kt_end) gload(kt + 1);
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    rd(1, S1{});
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    mm(S0{});
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    rd(2, S0{});
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    mm(S1{});
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    rd(3, S1{});
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    mm(S0{});
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    mm(S1{});
  }
}

template <int NS, bool PIN>
__global__ __launch_bounds__(256) void pv_kernel_v7(rmem_pv_args a) {
  using Cfg = GemmCfg<128, 128, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 127) / 128;
  const int nq = a.Npad / 128;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pl = j / nct;
  const int ctile = j - pl * nct;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps = a.Npad / 64;
  const int k_lo = 0, k_hi = a.T * tps;
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  int lo = k_lo + z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;
  PBlockedOperand lx{a.ph, a.pl, (long)a.Npad, qtile * 128};
  SlotLut lut;
  lut.load(a.slot_map, a.T);
  VtOperand ly{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols};
  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop_v7<Cfg, PIN>(f, lx, ly, lo, hi, smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float* out = a.part + (long)z * a.Npad * a.ncols;
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = ctile * 128 + frag_col<Cfg>(wc, tn, lane);
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 128 + frag_row<Cfg>(wr, tm, r, lane);
        out[(long)q * a.ncols + col] = f.acc[tm][tn][r];
      }
  }
}


// ---------------------------------------------------------------------------------------------
// V8 (timing only, results are NOT those of the shipped kernel): the shipped loop with P as ONE
// plane and V^T as two planes -- 2 MFMAs per product instead of 3, 3 staged planes instead of 4.
// What the "P as one fp16 plane" plan of tools/precision_study.py would cost on this kernel.
template <class Cfg, class LX, class LY>
__device__ __forceinline__ void gemm_mainloop_p1v2(GemmFrag<Cfg>& f, const LX& lx, const LY& ly, int kt_begin,
                                                   int kt_end, char* smem) {
  constexpr int XCH = Cfg::XCH, YCH = Cfg::YCH, TM = Cfg::TM, TN = Cfg::TN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (kt_begin >= kt_end) return;
  u32x4_t xr[XCH], yr[2 * YCH];
  auto gload = [&](int kt) __attribute__((always_inline)) {
    const TileView tx = lx.tile(kt);
    const TileView ty = ly.tile(kt);
    static_for<XCH>([&](auto I) {
      const int id = tid + I.value * Cfg::THREADS;
      xr[I.value] = *lx.ptr(tx, 0, id >> 3, id & 7);
    });
    static_for<2>([&](auto P) {
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        yr[P.value * YCH + I.value] = *ly.ptr(ty, P.value, id >> 3, id & 7);
      });
    });
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    static_for<XCH>([&](auto I) {
      const int id = tid + I.value * Cfg::THREADS;
      *reinterpret_cast<u32x4_t*>(smem + lds_swz(id >> 3, id & 7)) = xr[I.value];
    });
    static_for<2>([&](auto P) {
      char* yb = smem + Cfg::X_BYTES + P.value * Cfg::Y_BYTES;
      static_for<YCH>([&](auto I) {
        const int id = tid + I.value * Cfg::THREADS;
        *reinterpret_cast<u32x4_t*>(yb + lds_swz(id >> 3, id & 7)) = yr[P.value * YCH + I.value];
      });
    });
  };
  gload(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 1 < kt_end) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      bf16x8_t a[TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * Cfg::WM + i * 32 + (lane & 31);
        a[i] = *reinterpret_cast<const bf16x8_t*>(smem + lds_swz(row, chunk));
      }
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wc * Cfg::WN + j * 32 + (lane & 31);
          b[p][j] = *reinterpret_cast<const bf16x8_t*>(smem + Cfg::X_BYTES + p * Cfg::Y_BYTES + lds_swz(row, chunk));
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[1][j], f.acc[i][j], 0, 0, 0);
          f.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[0][j], f.acc[i][j], 0, 0, 0);
        }
    }
  }
}

__global__ __launch_bounds__(256) void pv_kernel_v8(rmem_pv_args a) {
  using Cfg = GemmCfg<128, 128, 3>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 127) / 128;
  const int nq = a.Npad / 128;
  const int npairs = nq * a.ksplits;
  const int chunk = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pl = j / nct;
  const int ctile = j - pl * nct;
  const int pair = xcd * chunk + pl;
  if (pl >= chunk || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps = a.Npad / 64;
  const int per = (a.T * tps + a.ksplits - 1) / a.ksplits;
  int lo = z * per, hi = lo + per;
  if (hi > a.T * tps) hi = a.T * tps;
  PBlockedOperand lx{a.ph, a.pl, (long)a.Npad, qtile * 128};
  SlotLut lut;
  lut.load(a.slot_map, a.T);
  VtOperand ly{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols};
  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop_p1v2<Cfg>(f, lx, ly, lo, hi, smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float* out = a.part + (long)z * a.Npad * a.ncols;
#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = ctile * 128 + frag_col<Cfg>(wc, tn, lane);
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qtile * 128 + frag_row<Cfg>(wr, tm, r, lane);
        out[(long)q * a.ncols + col] = f.acc[tm][tn][r];
      }
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
  const int N = 1674, Npad = 1792, T = 4, S = 6, ncols = 1024, ks = 4;
  const size_t p_elems = (size_t)T * Npad * Npad;            // blocked [key/32][Npad][32]
  const size_t v_elems = (size_t)S * ncols * Npad;
  std::vector<unsigned short> hp(p_elems), hv(v_elems);
  srand(1);
  for (auto& x : hp) x = 0x3c00 + (rand() & 0xff);   // small positive bf16 values
  for (auto& x : hv) x = 0x3f00 + (rand() & 0xff);
  bf16_t *ph, *pl, *vh, *vl;
  float* part;
  long long* stamps;
  int* smap;
  CK(hipMalloc(&ph, p_elems * 2)); CK(hipMalloc(&pl, p_elems * 2));
  CK(hipMalloc(&vh, v_elems * 2)); CK(hipMalloc(&vl, v_elems * 2));
  CK(hipMalloc(&part, (size_t)8 * Npad * ncols * 4));
  const int nsteps_max = 128;
  CK(hipMalloc(&stamps, 3 * nsteps_max * NSTAMP * 8));
  CK(hipMemset(stamps, 0, 3 * nsteps_max * NSTAMP * 8));
  CK(hipMalloc(&smap, 64));
  int hm[16] = {0, 1, 2, 3};
  CK(hipMemcpy(smap, hm, 64, hipMemcpyHostToDevice));
  CK(hipMemcpy(ph, hp.data(), p_elems * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(pl, hp.data(), p_elems * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(vh, hv.data(), v_elems * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(vl, hv.data(), v_elems * 2, hipMemcpyHostToDevice));
  rmem_pv_args a{};
  a.mode = 0; a.ph = ph; a.pl = pl; a.vh = vh; a.vl = vl; a.v_slot_stride = (long)ncols * Npad;
  a.slot_map = smap; a.T = T; a.N = N; a.Npad = Npad; a.ncols = ncols; a.h = 31; a.w = 54;
  a.part = part; a.ksplits = ks; a.nsplit = 3;
  using Cfg = GemmCfg<128, 128, 3>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_kernel_traced<3>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
  const int nct = 8, chunk = pv_chunk((Npad / 128) * ks);
  dim3 grid(8 * chunk * nct);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) rmem_attn_pv(&a, nullptr);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) rmem_attn_pv(&a, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("shipped pv_kernel<3>: %.2f us/launch (grid %d)\n", ms * 1e3 / 20, grid.x);
  {
    std::vector<float> ref((size_t)ks * Npad * ncols), got(ref.size());
    CK(hipMemcpy(ref.data(), part, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(part, 0, ref.size() * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_kernel_v2<3>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pv_kernel_v2<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), part, got.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < ref.size(); ++i) { maxd = fmax(maxd, fabs((double)ref[i] - got[i])); maxv = fmax(maxv, fabs((double)ref[i])); }
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pv_kernel_v2<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("V2 (32-key steps, LDS double buffer, 1 barrier/step): %.2f us/launch; max |diff| vs shipped %.3g (max |ref| %.3g)\n", ms * 1e3 / 20, maxd, maxv);
  }
  {
    std::vector<float> ref((size_t)ks * Npad * ncols), got(ref.size());
    for (int i = 0; i < 2; ++i) rmem_attn_pv(&a, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), part, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(part, 0, ref.size() * 4));
    constexpr int LDS3 = 3 * 2 * (256 * 64 + 128 * 64);
    const int chunk3 = pv_chunk((Npad / 256) * ks);
    dim3 grid3(8 * chunk3 * nct);
    auto run3 = [&](auto kern, const char* tag) {
      CK(hipMemset(part, 0, ref.size() * 4));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS3));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid3, dim3(512), LDS3, 0, a, (long long*)nullptr);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), part, got.size() * 4, hipMemcpyDeviceToHost));
      double maxd = 0, maxv = 0;
      for (size_t i = 0; i < ref.size(); ++i) { maxd = fmax(maxd, fabs((double)ref[i] - got[i])); maxv = fmax(maxv, fabs((double)ref[i])); }
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid3, dim3(512), LDS3, 0, a, (long long*)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms);
      }
      printf("V3 %s (grid %d): %.2f us/launch; max |diff| vs shipped %.3g (max |ref| %.3g)\n", tag, grid3.x, best * 1e3 / 20, maxd, maxv);
      if (getenv("STAMPS")) {
        long long* st3; CK(hipMalloc(&st3, 8 * 128 * 4 * 8)); CK(hipMemset(st3, 0, 8 * 128 * 4 * 8));
        hipLaunchKernelGGL(kern, grid3, dim3(512), LDS3, 0, a, st3);
        CK(hipDeviceSynchronize());
        std::vector<long long> h3(8 * 128 * 4);
        CK(hipMemcpy(h3.data(), st3, h3.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < 8; w += 3) {
          long long* q = h3.data() + w * 128 * 4;
          double sw = 0, sb = 0, sc = 0, gap = 0; int n = 0;
          for (int k = 1; k < 128 && q[k * 4]; ++k, ++n) {
            sw += q[k * 4 + 1] - q[k * 4]; sb += q[k * 4 + 2] - q[k * 4 + 1]; sc += q[k * 4 + 3] - q[k * 4 + 2]; gap += q[k * 4] - q[(k - 1) * 4 + 3];
          }
          printf("    wave %d: %d steps, mean ticks: vmcnt-wait %.0f barrier %.0f issue+compute %.0f loop-overhead %.0f; total %lld\n", w, n, sw / n, sb / n, sc / n, gap / n, q[n * 4 + 3] - q[0]);
        }
      }
    };
    {
      constexpr int LDS5 = 4 * 2 * (256 * 32 + 256 * 32);
      const int nct5 = (ncols + 255) / 256;
      auto run5 = [&](auto kern, const char* tag, int ks5) {
        rmem_pv_args b = a; b.ksplits = ks5;
        const int chunk5 = pv_chunk(((Npad + 255) / 256) * ks5);
        dim3 grid5(8 * chunk5 * nct5);
        CK(hipMemset(part, 0, (size_t)8 * Npad * ncols * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS5));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid5, dim3(512), LDS5, 0, b, (long long*)nullptr);
        CK(hipDeviceSynchronize());
        // compare the sum over splits (split boundaries differ from the shipped kernel's)
        std::vector<float> g5((size_t)ks5 * Npad * ncols);
        CK(hipMemcpy(g5.data(), part, g5.size() * 4, hipMemcpyDeviceToHost));
        double maxd = 0, maxv = 0;
        for (size_t i = 0; i < (size_t)N * ncols; ++i) {
          double r0 = 0, r1 = 0;
          for (int zz = 0; zz < ks; ++zz) r0 += ref[(size_t)zz * Npad * ncols + i];
          for (int zz = 0; zz < ks5; ++zz) r1 += g5[(size_t)zz * Npad * ncols + i];
          maxd = fmax(maxd, fabs(r0 - r1)); maxv = fmax(maxv, fabs(r0));
        }
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid5, dim3(512), LDS5, 0, b, (long long*)nullptr);
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
          best = fminf(best, ms);
        }
        printf("V5 %s ks=%d (grid %d): %.2f us/launch; max |diff of split sums| vs shipped %.3g (max |ref| %.3g)\n", tag, ks5, grid5.x, best * 1e3 / 20, maxd, maxv);
        if (getenv("STAMPS")) {
          long long* st3; CK(hipMalloc(&st3, 8 * 128 * 4 * 8)); CK(hipMemset(st3, 0, 8 * 128 * 4 * 8));
          hipLaunchKernelGGL(kern, grid5, dim3(512), LDS5, 0, b, st3);
          CK(hipDeviceSynchronize());
          std::vector<long long> h3(8 * 128 * 4);
          CK(hipMemcpy(h3.data(), st3, h3.size() * 8, hipMemcpyDeviceToHost));
          for (int w = 0; w < 8; w += 4) {
            long long* q = h3.data() + w * 128 * 4;
            double sw = 0, sb = 0, sc = 0, gap = 0; int n = 0;
            for (int k = 1; k < 127 && q[k * 4]; ++k, ++n) {
              sw += q[k * 4 + 1] - q[k * 4]; sb += q[k * 4 + 2] - q[k * 4 + 1]; sc += q[k * 4 + 3] - q[k * 4 + 2]; gap += q[k * 4] - q[(k - 1) * 4 + 3];
            }
            printf("    wave %d: %d steps, mean ticks: s0->s1 %.0f s1->s2 %.0f s2->s3 %.0f loop %.0f; total %lld\n", w, n, sw / n, sb / n, sc / n, gap / n, q[n * 4 + 3] - q[0]);
          }
        }
      };
      run5(&pv_kernel_v5<3, false>, "plain (1 barrier/step)", 8);
      run5(&pv_kernel_v5<3, true>, "staggered wave groups (2 barriers/step)", 8);
      run5(&pv_kernel_v5<3, true>, "staggered wave groups (2 barriers/step)", 4);
    }
    {
      constexpr int LDS6 = 2 * Cfg::LDS_BYTES;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_kernel_v6<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS6));
      for (int ks6 = 4; ks6 <= 8; ks6 += 4) {
        rmem_pv_args b = a; b.ksplits = ks6;
        const int chunk6 = pv_chunk((Npad / 128) * ks6);
        dim3 grid6(8 * chunk6 * (nct / 2));
        CK(hipMemset(part, 0, (size_t)8 * Npad * ncols * 4));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pv_kernel_v6<3>), grid6, dim3(512), LDS6, 0, b);
        CK(hipDeviceSynchronize());
        std::vector<float> g6((size_t)ks6 * Npad * ncols);
        CK(hipMemcpy(g6.data(), part, g6.size() * 4, hipMemcpyDeviceToHost));
        double maxd = 0, maxv = 0;
        for (size_t i = 0; i < (size_t)N * ncols; ++i) {
          double r0 = 0, r1 = 0;
          for (int zz = 0; zz < ks; ++zz) r0 += ref[(size_t)zz * Npad * ncols + i];
          for (int zz = 0; zz < ks6; ++zz) r1 += g6[(size_t)zz * Npad * ncols + i];
          maxd = fmax(maxd, fabs(r0 - r1)); maxv = fmax(maxv, fabs(r0));
        }
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pv_kernel_v6<3>), grid6, dim3(512), LDS6, 0, b);
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
          best = fminf(best, ms);
        }
        printf("V6 (two anti-phased 128x128 groups per workgroup) ks=%d (grid %d): %.2f us/launch; max |diff of split sums| vs shipped %.3g (max |ref| %.3g)\n", ks6, grid6.x, best * 1e3 / 20, maxd, maxv);
      }
    }
    {
      auto run7 = [&](auto kern, const char* tag) {
        CK(hipMemset(part, 0, ref.size() * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), part, got.size() * 4, hipMemcpyDeviceToHost));
        double maxd = 0;
        for (size_t i = 0; i < ref.size(); ++i) maxd = fmax(maxd, fabs((double)ref[i] - got[i]));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, 0, a);
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
          best = fminf(best, ms);
        }
        printf("V7 %s: %.2f us/launch; max |diff| vs shipped %.3g\n", tag, best * 1e3 / 20, maxd);
      };
      run7(&pv_kernel_v7<3, false>, "fragment double buffer (compiler free to move)");
      run7(&pv_kernel_v7<3, true>, "fragment double buffer (order pinned with sched_barrier)");
    }
    {
      constexpr int LDS8 = 3 * 128 * 128;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_kernel_v8), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pv_kernel_v8, grid, dim3(256), LDS8, 0, a);
      CK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pv_kernel_v8, grid, dim3(256), LDS8, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms);
      }
      printf("V8 (timing only: P one plane, V^T two planes, 2 MFMAs per product, 48 KB LDS): %.2f us/launch\n", best * 1e3 / 20);
    }
    run3(&pv_kernel_v3<3, 0>, "var0 (compiler order)");
    run3(&pv_kernel_v3<3, 1>, "var1 (fragments read up front, then DMA issue, then MFMAs)");
    run3(&pv_kernel_v3<3, 2>, "var2 (var1 + s_setprio around the MFMAs)");
  }
  for (int pattern = 9; pattern <= 12; ++pattern) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pv_kernel_traced<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a, (long long*)nullptr, nsteps_max, 0, pattern);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pv_kernel_traced<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a, (long long*)nullptr, nsteps_max, 0, pattern);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      best = fminf(best, ms);
    }
    printf("shipped structure, k-order rotation pattern %d (9 = none, 10 = by column tile, 11 = by column+query tile, 12 = by block): %.2f us/launch\n", pattern, best * 1e3 / 20);
  }
  if (getenv("SWEEP"))
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int delay = 0; delay <= 6; ++delay) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pv_kernel_traced<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a, (long long*)nullptr, nsteps_max, delay, pattern);
      CK(hipDeviceSynchronize());
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pv_kernel_traced<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a, (long long*)nullptr, nsteps_max, delay, pattern);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("pattern %d delay %d x 1024 cycles: %.2f us/launch\n", pattern, delay, ms * 1e3 / 20);
    }
  int best_delay = getenv("DELAY") ? atoi(getenv("DELAY")) : 3;
  hipLaunchKernelGGL((pv_kernel_traced<3>), grid, dim3(256), Cfg::LDS_BYTES, 0, a, stamps, nsteps_max, best_delay, 0);
  CK(hipDeviceSynchronize());
  std::vector<long long> hs(3 * nsteps_max * NSTAMP);
  CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  const char* names[NSTAMP] = {"top", "bar1", "vmcnt", "lstore", "bar2", "compute"};
  for (int b = 0; b < 3; ++b) {
    long long* s = hs.data() + b * nsteps_max * NSTAMP;
    printf("block %d: stamps relative to loop start (clock64 ticks); per step: ", b);
    for (int i = 0; i < NSTAMP; ++i) printf("%s ", names[i]);
    printf("\n");
    double sum[NSTAMP] = {0};
    int n = 0;
    for (int k = 0; k < nsteps_max && s[k * NSTAMP]; ++k, ++n) {
      long long prev = k ? s[(k - 1) * NSTAMP + 5] : s[0];
      if (k < 6) printf("  step %2d: +%lld |", k, s[k * NSTAMP] - s[0]);
      for (int i = 0; i < NSTAMP; ++i) {
        long long d = s[k * NSTAMP + i] - (i ? s[k * NSTAMP + i - 1] : prev);
        if (k < 6) printf(" %6lld", d);
        if (k) sum[i] += d;
      }
      if (k < 6) printf("\n");
    }
    printf("  mean over steps 1..%d:", n - 1);
    for (int i = 0; i < NSTAMP; ++i) printf(" %s=%.0f", names[i], sum[i] / (n - 1));
    printf("  total loop %lld ticks for %d steps\n", s[(n - 1) * NSTAMP + 5] - s[0], n);
  }
  return 0;
}
