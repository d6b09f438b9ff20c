// gemm_core.h -- LDS-tiled split-fp16 MFMA main loop for gfx950 (wave64).
//
//   D[i][j] = sum_k X[i][k] * Y[j][k]        (both operands reduction-contiguous)
//
// Block = 256 threads = 4 waves arranged 2x2 over a BM x BN tile; every wave owns a
// (BM/2) x (BN/2) sub-tile made of 32x32 MFMA tiles (v_mfma_f32_32x32x16_f16, fp32
// accumulate).  K is consumed in steps of BK = 64 fp16 (one 128-byte row per tile row).
//
// LDS image of one operand plane: row r occupies bytes [128 r, 128 r + 128); the eight
// 16-byte chunks of a row are stored at chunk position  c ^ ((r >> 1) & 7).  With
// row = lane this makes the ds_read_b128 fragment reads conflict-free (the 16 lanes of
// every b128 service group hit 16 distinct 16-byte slots of the 256-byte bank row) and
// the ds_write_b128 staging writes (8 consecutive lanes = 8 chunks of one row) too.
//
// Pipeline: global -> registers (issued one k-tile ahead, T14 "issue early / write late")
// -> LDS -> fragments.  Two __syncthreads per k-tile.
//
// Fragment layout (MI355X guide section 3): operand lane l holds row (l & 31), k-group
// (l >> 5) (8 consecutive k); accumulator register r of lane l is
// D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
#pragma once
#include "rmem_common.h"
#include <type_traits>

template <int BM_, int BN_, int NSPLIT_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, NSPLIT = NSPLIT_;
  static constexpr int BK = 64;
  static constexpr int THREADS = 256;
  static constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int NPL = (NSPLIT == 1) ? 1 : 2;   // planes per operand
  static constexpr int XCH = BM * 8 / THREADS;        // 16-byte chunks per thread per plane
  static constexpr int YCH = BN * 8 / THREADS;
  static constexpr int X_BYTES = BM * 128, Y_BYTES = BN * 128;
  static constexpr int LDS_BYTES = NPL * (X_BYTES + Y_BYTES);
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tile must be a multiple of 64");
};

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

__device__ __forceinline__ int lds_swz(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Per-k-tile view of an operand: everything that is uniform over the tile (segment
// choice, slot lookup, divisions) is resolved ONCE per k-tile in tile(); ptr() is then
// pure per-lane address arithmetic, so the 8-16 staging loads of a tile issue back to back.
struct TileView {
  const h16_t* hi;
  const h16_t* lo;
  long ld;
};

// Row-major operand [rows][ld] (fp16 hi/lo planes), optionally continued by a second source
// for k-tiles >= kt_split (used to concatenate two activations along K).
struct RowMajorOperand {
  const h16_t* hi;
  const h16_t* lo;
  long ld;
  const h16_t* hi2;
  const h16_t* lo2;
  long ld2;
  int kt_split;   // number of k-tiles served by the first source
  int row0;       // first row of this block's tile
  int rows;       // valid rows (loads are clamped to rows-1)
  __device__ __forceinline__ TileView tile(int kt) const {
    if (kt < kt_split) return TileView{hi + kt * 64, lo ? lo + kt * 64 : nullptr, ld};
    return TileView{hi2 + (kt - kt_split) * 64, lo2 ? lo2 + (kt - kt_split) * 64 : nullptr, ld2};
  }
  __device__ __forceinline__ const u32x4_t* ptr(const TileView& t, int plane, int r, int c) const {
    int gr = row0 + r;
    gr = gr < rows ? gr : rows - 1;
    const h16_t* b = plane ? t.lo : t.hi;
    return reinterpret_cast<const u32x4_t*>(b + (long)gr * t.ld + c * 8);
  }
};

template <class Cfg>
struct GemmFrag {
  f32x16_t acc[Cfg::TM][Cfg::TN];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
};

// row (within the block tile) of accumulator register r
template <class Cfg>
__device__ __forceinline__ int frag_row(int wr, int tm, int r, int lane) {
  return wr * Cfg::WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <class Cfg>
__device__ __forceinline__ int frag_col(int wc, int tn, int lane) {
  return wc * Cfg::WN + tn * 32 + (lane & 31);
}

template <class Cfg, class LX, class LY>
__device__ __forceinline__ void gemm_mainloop(GemmFrag<Cfg>& f, const LX& lx, const LY& ly,
                                              int kt_begin, int kt_end, char* smem) {
  constexpr int NPL = Cfg::NPL, XCH = Cfg::XCH, YCH = Cfg::YCH, TM = Cfg::TM, TN = Cfg::TN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (kt_begin >= kt_end) return;

  // Staging registers.  Indexed only through static_for (compile-time indices) so the
  // arrays are promoted to VGPRs (a rolled loop here puts them in scratch).
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

  gload(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();  // all waves finished reading the previous tile
    lstore();
    __syncthreads();
    if (kt + 1 < kt_end) gload(kt + 1);  // in flight while this tile is multiplied
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      frag8_t a[NPL][TM], b[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const char* xb = smem + p * Cfg::X_BYTES;
        const char* yb = smem + NPL * Cfg::X_BYTES + p * Cfg::Y_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * Cfg::WM + i * 32 + (lane & 31);
          a[p][i] = *reinterpret_cast<const frag8_t*>(xb + lds_swz(row, chunk));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wc * Cfg::WN + j * 32 + (lane & 31);
          b[p][j] = *reinterpret_cast<const frag8_t*>(yb + lds_swz(row, chunk));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (Cfg::NSPLIT == 3) {  // small terms first
            f.acc[i][j] = RMEM_MFMA(a[0][i], b[1][j], f.acc[i][j]);
            f.acc[i][j] = RMEM_MFMA(a[1][i], b[0][j], f.acc[i][j]);
          }
          f.acc[i][j] = RMEM_MFMA(a[0][i], b[0][j], f.acc[i][j]);
        }
    }
  }
}
