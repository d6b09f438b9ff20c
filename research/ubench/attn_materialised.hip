// attn.hip -- memory-read attention (long-term bank / windowed short-term / self) as
// three MFMA launches over a materialised split-fp16 probability matrix.
// See include/rmem_hip.h for the contract and DESIGN.md for the roofline accounting.
#include "../../include/rmem_hip.h"
#include "gemm_core.h"
#include "launch.h"
#include "attn_common.h"

// ------------------------------------------------------------------ scores (swapped: rows = keys)
template <int NS, int PASS>
__device__ __forceinline__ void scores_body(const rmem_scores_args& a, int bx, int by, char* smem) {
  using Cfg = GemmCfg<128, 128, NS>;
  const int qtile = by;
  const int tiles_per_slot = a.Npad / 128;
  int t = 0, ktile;  // ktile: key tile inside the slot
  if (a.mode == 0) {
    t = bx / tiles_per_slot;
    ktile = bx - t * tiles_per_slot;
  } else {
    int t_lo, t_hi;
    band_tiles(qtile, a.N, a.h, a.w, t_lo, t_hi);
    ktile = t_lo + bx;
    if (ktile >= t_hi) return;
  }
  const int phys = a.slot_map ? a.slot_map[t] : t;

  RowMajorOperand lx, ly;
  lx.hi = a.kh + (long)phys * a.k_slot_stride;
  lx.lo = a.kl ? a.kl + (long)phys * a.k_slot_stride : nullptr;
  lx.ld = 128;
  lx.hi2 = lx.lo2 = nullptr;
  lx.ld2 = 0;
  lx.kt_split = 1 << 30;
  lx.row0 = ktile * 128;
  lx.rows = a.Npad;
  ly.hi = a.qh;
  ly.lo = a.ql;
  ly.ld = 128;
  ly.hi2 = ly.lo2 = nullptr;
  ly.ld2 = 0;
  ly.kt_split = 1 << 30;
  ly.row0 = qtile * 128;
  ly.rows = a.Npad;

  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop<Cfg>(f, lx, ly, 0, 2, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const float inv_w = a.mode == 1 ? 1.0f / (float)a.w : 0.f;
  // split precision without a low plane: P is ONE fp16 plane (11 significant bits; P.V then takes
  // 2 MFMAs per product and 3 staged planes instead of 3 and 4, see pv16_kernel)
  const bool p16 = PASS == 1 && NS == 3 && a.pl == nullptr;
  // key axis of P / lpart: logical slot-major
  const long key_axis0 = (long)t * a.Npad + ktile * 128 + wr * 64;

#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int q = qtile * 128 + frag_col<Cfg>(wc, tn, lane);
    const bool qvalid = q < a.N;
    float bias_q = 0.f;
    int qy = 0, qx = 0;
    const float* Rq = nullptr;
    if (a.mode == 0) {
      if (a.bias && qvalid) bias_q = a.bias[(long)q * a.T + t];
    } else {
      qy = fast_div(q, inv_w);
      qx = q - qy * a.w;
      Rq = a.R + (long)(qvalid ? q : 0) * a.ldr;
    }
    float mrow = 0.f;
    if (PASS == 1) mrow = qvalid ? dec_ordered(a.rowmax[q]) : 0.f;
    float mx = -3.0e38f, lsum = 0.f;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tok = ktile * 128 + frag_row<Cfg>(wr, tm, r, lane);
        bool valid = qvalid && tok < a.N;
        float s;
        if (a.mode == 0) {
          s = a.scale * (f.acc[tm][tn][r] + bias_q);
        } else {
          const int ky = fast_div(tok, inv_w);
          const int kx = tok - ky * a.w;
          const int dy = ky - qy, dx = kx - qx;
          valid = valid && dy >= -7 && dy <= 7 && dx >= -7 && dx <= 7;
          const float rb = valid ? Rq[(dy + 7) * 15 + dx + 7] : 0.f;
          s = a.scale * f.acc[tm][tn][r] + rb;
        }
        if (PASS == 0) {
          if (valid) mx = fmaxf(mx, s);
        } else {
          float p = valid ? exp_weight(s - mrow) : 0.f;
          if (p16) p = h_bits2f(f2h_bits(p));   // the row sum is taken over the STORED weights: the rounding of a dominant weight cancels in the normalisation
          pv[r] = p;
          lsum += p;
        }
      }
      if (PASS == 1) {
        // P blocked [key/32][Npad][32]: this lane owns keys 8g + 4*(lane>>5) + {0..3}
        const long kb = (key_axis0 + tm * 32) >> 5;
        const long base = (kb * a.Npad + q) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h16_t hi[4], lo[4];
          if (p16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) hi[e] = f2h_bits(pv[4 * g + e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) split_f16(pv[4 * g + e], hi[e], lo[e]);
          }
          uint2 vh, vl;
          vh.x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
          vh.y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
          *reinterpret_cast<uint2*>(a.ph + base + 8 * g) = vh;
          if (a.pl) {
            vl.x = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
            vl.y = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
            *reinterpret_cast<uint2*>(a.pl + base + 8 * g) = vl;
          }
        }
      }
    }
    if (PASS == 0) {
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (lane < 32 && qvalid && mx > -2.9e38f) atomicMax(a.rowmax + q, enc_ordered(mx));
    } else {
      lsum += __shfl_xor(lsum, 32);
      if (lane < 32) a.lpart[(long)q * a.nparts + (key_axis0 >> 6)] = lsum;
    }
  }
}

template <int NS, int PASS>
__global__ __launch_bounds__(256) void scores_kernel(rmem_scores_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  scores_body<NS, PASS>(a, blockIdx.x, blockIdx.y, smem);
}

// Two independent reads of the same frame (long-term bank + windowed short-term) in one launch:
// blocks [0, na) serve problem a (x-extent gxa), the rest problem b (x-extent gxb).
template <int NS, int PASS>
__global__ __launch_bounds__(256) void scores2_kernel(rmem_scores_args a, rmem_scores_args b, int na, int gxa, int gxb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int i = blockIdx.x;
  if (i < na) scores_body<NS, PASS>(a, i % gxa, i / gxa, smem);
  else scores_body<NS, PASS>(b, (i - na) % gxb, (i - na) / gxb, smem);
}

static int max_band_tiles(int N, int Npad, int h, int w) {
  int mx = 0;
  for (int qt = 0; qt < Npad / 128; ++qt) {
    int lo, hi;
    band_tiles(qt, N, h, w, lo, hi);
    if (hi - lo > mx) mx = hi - lo;
  }
  return mx;
}

template <int NS, int PASS>
static int launch_scores(const rmem_scores_args& a, hipStream_t s) {
  using Cfg = GemmCfg<128, 128, NS>;
  const int qtiles = a.Npad / 128;
  const int ktiles = a.mode == 0 ? a.T * (a.Npad / 128) : max_band_tiles(a.N, a.Npad, a.h, a.w);
  // per launch: the attribute belongs to the (device, function) pair, and a cached "already set"
  // flag would be process-wide state shared by every device and host thread
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scores_kernel<NS, PASS>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  hipLaunchKernelGGL((scores_kernel<NS, PASS>), dim3(ktiles, qtiles), dim3(256), Cfg::LDS_BYTES, s, a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

static int scores_args_ok(const rmem_scores_args& a) {
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) != 0 || a.T <= 0) return 0;
  if (!a.kh || !a.qh || !a.rowmax) return 0;
  if (a.mode == 1 && (!a.R || a.h * a.w != a.N || a.T != 1)) return 0;
  if (a.mode != 0 && a.mode != 1) return 0;
  if (a.pass == 1 && (!a.ph || !a.lpart)) return 0;
  if (a.pass == 1 && a.nsplit == 3 && (!a.kl || !a.ql)) return 0;      // pl == NULL: P as one fp16 plane
  return 1;
}

template <int NS, int PASS>
static int launch_scores2(const rmem_scores_args& a, const rmem_scores_args& b, hipStream_t s) {
  using Cfg = GemmCfg<128, 128, NS>;
  const int gxa = a.mode == 0 ? a.T * (a.Npad / 128) : max_band_tiles(a.N, a.Npad, a.h, a.w);
  const int gxb = b.mode == 0 ? b.T * (b.Npad / 128) : max_band_tiles(b.N, b.Npad, b.h, b.w);
  const int na = gxa * (a.Npad / 128), nb = gxb * (b.Npad / 128);
  // per launch: the attribute belongs to the (device, function) pair, and a cached "already set"
  // flag would be process-wide state shared by every device and host thread
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scores2_kernel<NS, PASS>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  hipLaunchKernelGGL((scores2_kernel<NS, PASS>), dim3(na + nb), dim3(256), Cfg::LDS_BYTES, s, a, b, na, gxa, gxb);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_attn_scores2(const rmem_scores_args* ap, const rmem_scores_args* bp, void* stream) {
  if (!ap || !bp) return RMEM_ERR_INVALID;
  const rmem_scores_args& a = *ap;
  const rmem_scores_args& b = *bp;
  if (!scores_args_ok(a) || !scores_args_ok(b) || a.pass != b.pass || a.nsplit != b.nsplit) return RMEM_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.pass == 0) return launch_scores2<1, 0>(a, b, s);
  if (a.pass != 1) return RMEM_ERR_INVALID;
  if (a.nsplit == 3) return launch_scores2<3, 1>(a, b, s);
  if (a.nsplit == 1) return launch_scores2<1, 1>(a, b, s);
  return RMEM_ERR_INVALID;
}

extern "C" int rmem_attn_scores(const rmem_scores_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  const rmem_scores_args& a = *ap;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) != 0 || a.T <= 0) return RMEM_ERR_INVALID;
  if (!a.kh || !a.qh || !a.rowmax) return RMEM_ERR_INVALID;
  if (a.mode == 1 && (!a.R || a.h * a.w != a.N || a.T != 1)) return RMEM_ERR_INVALID;
  if (a.mode != 0 && a.mode != 1) return RMEM_ERR_INVALID;
  if (a.pass == 0) return launch_scores<1, 0>(a, s);
  if (a.pass != 1 || !a.ph || !a.lpart) return RMEM_ERR_INVALID;
  if (a.nsplit == 3) {
    if (!a.kl || !a.ql) return RMEM_ERR_INVALID;       // pl == NULL: P as one fp16 plane
    return launch_scores<3, 1>(a, s);
  }
  if (a.nsplit == 1) return launch_scores<1, 1>(a, s);
  return RMEM_ERR_INVALID;
}

// ------------------------------------------------------------------ P . V
struct PBlockedOperand {
  const h16_t* hi;
  const h16_t* lo;
  long npad;
  int row0;
  __device__ __forceinline__ TileView tile(int kt) const {
    const long off = (long)kt * 2 * npad * 32;
    return TileView{hi + off, lo ? lo + off : nullptr, npad * 32};   // ld = one 32-key block
  }
  __device__ __forceinline__ const u32x4_t* ptr(const TileView& t, int plane, int r, int c) const {
    const h16_t* b = plane ? t.lo : t.hi;
    long q = row0 + r;
    q = q < npad ? q : npad - 1;
    return reinterpret_cast<const u32x4_t*>(b + (c >> 2) * t.ld + q * 32 + (c & 3) * 8);
  }
};

struct VtOperand {
  const h16_t* hi;
  const h16_t* lo;
  long slot_stride, ld;
  SlotLut lut;
  int tps;  // 64-key tiles per slot
  int row0, rows;
  __device__ __forceinline__ TileView tile(int kt) const {
    const int t = kt / tps;
    const int phys = lut(t);
    const long off = (long)phys * slot_stride + (kt - t * tps) * 64;
    return TileView{hi + off, lo ? lo + off : nullptr, ld};
  }
  __device__ __forceinline__ const u32x4_t* ptr(const TileView& t, int plane, int r, int c) const {
    int j = row0 + r;
    j = j < rows ? j : rows - 1;
    const h16_t* b = plane ? t.lo : t.hi;
    return reinterpret_cast<const u32x4_t*>(b + (long)j * t.ld + c * 8);
  }
};

// k-tile index over the VALID 64-key tiles of the bank (tv per slot) -> actual tile index (tps per
// slot): the tiles of a slot that hold only padding keys (P is exactly 0 there) are never visited.
template <class Inner>
struct SkipPadTiles {
  Inner in;
  int tv, tps;
  __device__ __forceinline__ TileView tile(int i) const {
    const int t = i / tv;
    return in.tile(t * tps + (i - t * tv));
  }
  __device__ __forceinline__ const u32x4_t* ptr(const TileView& t, int plane, int r, int c) const {
    return in.ptr(t, plane, r, c);
  }
};

// Work decomposition of P.V: unit = (query tile, key split, column tile).  Units are
// laid out so that the 8 XCDs (block b runs on XCD b % 8 -- observed placement, used for
// speed only) each own a contiguous chunk of (split, query tile) pairs, split-major, and
// the ncols/128 column tiles of a pair are co-resident on that XCD: the P tile of a pair is
// fetched from the Infinity Cache once and hit in the XCD's L2 by the other column tiles,
// and pairs of one XCD share a key split, i.e. the same V^T key range.
__host__ __device__ inline int pv_chunk(int npairs) { return (npairs + 7) / 8; }

template <int NS>
__global__ __launch_bounds__(256) void pv_kernel(rmem_pv_args a) {
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
  const int tv = a.mode == 0 ? (a.N + 63) / 64 : tps;   // bank read: skip the all-padding tiles of every slot
  int k_lo, k_hi;
  if (a.mode == 0) {
    k_lo = 0;
    k_hi = a.T * tv;
  } else {
    int t_lo, t_hi;
    band_tiles(qtile, a.N, a.h, a.w, t_lo, t_hi);
    k_lo = 2 * t_lo;
    k_hi = 2 * t_hi;
  }
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  int lo = k_lo + z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;

  SlotLut lut;
  lut.load(a.slot_map, a.T);
  SkipPadTiles<PBlockedOperand> lx{PBlockedOperand{a.ph, a.pl, (long)a.Npad, qtile * 128}, tv, tps};
  SkipPadTiles<VtOperand> ly{VtOperand{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols},
                             tv, tps};

  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop<Cfg>(f, lx, ly, lo, hi, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

template <int NS>
static int launch_pv(const rmem_pv_args& a, hipStream_t s) {
  using Cfg = GemmCfg<128, 128, NS>;
  // per launch: the attribute belongs to the (device, function) pair, and a cached "already set"
  // flag would be process-wide state shared by every device and host thread
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_kernel<NS>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  const int nct = (a.ncols + 127) / 128;
  const int chunk = pv_chunk((a.Npad / 128) * a.ksplits);
  dim3 grid(8 * chunk * nct);
  hipLaunchKernelGGL((pv_kernel<NS>), grid, dim3(256), Cfg::LDS_BYTES, s, a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ P . V with P as ONE fp16 plane
// Bank read (mode 0) only.  P [one fp16 plane] x V^T [fp16 hi/lo planes]:  O = P.Vhi + P.Vlo,
// 2 MFMAs (v_mfma_f32_32x32x16_f16) per product and 48 KB of LDS per block instead of 3 and 64 KB.  P carries 11 significant bits; what that costs in label
// maps is measured in tools/precision_study.py (p16@long,self) and DESIGN.md section 3.
__global__ __launch_bounds__(256) void pv16_kernel(rmem_pv_args a) {
  using Cfg = GemmCfg<128, 128, 3>;
  constexpr int XCH = Cfg::XCH, YCH = Cfg::YCH, TM = Cfg::TM, TN = Cfg::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nct = (a.ncols + 127) / 128;
  const int nq = a.Npad / 128;
  const int npairs = nq * a.ksplits;
  const int chunk_ = pv_chunk(npairs);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int pl = j / nct;
  const int ctile = j - pl * nct;
  const int pair = xcd * chunk_ + pl;
  if (pl >= chunk_ || pair >= npairs) return;
  const int z = pair / nq;
  const int qtile = pair - z * nq;
  const int tps = a.Npad / 64;
  const int tv = (a.N + 63) / 64;
  const int k_hi = a.T * tv;
  const int per = (k_hi + a.ksplits - 1) / a.ksplits;
  int lo = z * per, hi = lo + per;
  if (hi > k_hi) hi = k_hi;

  SlotLut lut;
  lut.load(a.slot_map, a.T);
  SkipPadTiles<PBlockedOperand> lx{PBlockedOperand{a.ph, nullptr, (long)a.Npad, qtile * 128}, tv, tps};
  SkipPadTiles<VtOperand> ly{VtOperand{a.vh, a.vl, (long)a.v_slot_stride, (long)a.Npad, lut, tps, ctile * 128, a.ncols},
                             tv, tps};
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  GemmFrag<Cfg> f;
  f.zero();
  if (lo < hi) {
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
    gload(lo);
    for (int kt = lo; kt < hi; ++kt) {
      __syncthreads();
      lstore();
      __syncthreads();
      if (kt + 1 < hi) gload(kt + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = ks * 2 + (lane >> 5);
        f16x8_t pa[TM], vb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * Cfg::WM + i * 32 + (lane & 31);
          pa[i] = *reinterpret_cast<const f16x8_t*>(smem + lds_swz(row, chunk));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) {
            const int row = wc * Cfg::WN + jj * 32 + (lane & 31);
            vb[p][jj] = *reinterpret_cast<const f16x8_t*>(smem + Cfg::X_BYTES + p * Cfg::Y_BYTES + lds_swz(row, chunk));
          }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) {   // small term first
            f.acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[i], vb[1][jj], f.acc[i][jj], 0, 0, 0);
            f.acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[i], vb[0][jj], f.acc[i][jj], 0, 0, 0);
          }
      }
    }
  }
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

static int launch_pv16(const rmem_pv_args& a, hipStream_t s) {
  constexpr int LDS = 3 * 128 * 128;
  // per launch: the attribute belongs to the (device, function) pair, and a cached "already set"
  // flag would be process-wide state shared by every device and host thread
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pv16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const int nct = (a.ncols + 127) / 128;
  const int chunk = pv_chunk((a.Npad / 128) * a.ksplits);
  hipLaunchKernelGGL(pv16_kernel, dim3(8 * chunk * nct), dim3(256), LDS, s, a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_attn_pv(const rmem_pv_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  const rmem_pv_args& a = *ap;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) != 0 || a.T <= 0 || a.T > 16 || a.ksplits <= 0) return RMEM_ERR_INVALID;
  if (!a.ph || !a.vh || !a.part || a.ncols <= 0) return RMEM_ERR_INVALID;
  if (a.mode == 1 && (a.h * a.w != a.N || a.T != 1)) return RMEM_ERR_INVALID;
  if (a.nsplit == 3 && !a.vl) return RMEM_ERR_INVALID;
  if (a.nsplit != 1 && a.nsplit != 3) return RMEM_ERR_INVALID;
  if (a.nsplit == 3 && !a.pl) {            // P as one fp16 plane (written by rmem_attn_scores with pl == NULL)
    if (a.mode != 0) return RMEM_ERR_INVALID;
    return launch_pv16(a, s);
  }
  if (a.nsplit == 3) return launch_pv<3>(a, s);
  return launch_pv<1>(a, s);
}

// ------------------------------------------------------------------ combine + gate (+ mass)
__device__ __forceinline__ void combine_body(const rmem_combine_args& a, int q, float* slot_sum) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tps = a.Npad / 64;
  const float* lp = a.lpart + (long)q * a.nparts;
  int nslots = a.T;
  for (int t = wave; t < nslots; t += 4) {
    int p_lo = t * tps, p_hi = (t + 1) * tps;
    if (a.mode == 1) {
      int t_lo, t_hi;
      band_tiles(q / 128, a.N, a.h, a.w, t_lo, t_hi);
      p_lo = 2 * t_lo;
      p_hi = 2 * t_hi;
    }
    float s = 0.f;
    for (int p = p_lo + lane; p < p_hi; p += 64) s += lp[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) slot_sum[t] = s;
  }
  __syncthreads();
  float l = 0.f;
  for (int t = 0; t < nslots; ++t) l += slot_sum[t];
  const float inv_l = 1.0f / l;
  if (a.mass && tid < nslots) a.mass[(long)q * a.T + tid] = slot_sum[tid] * inv_l;

  for (int c = tid * 4; c < a.ncols; c += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < a.ksplits; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(a.part + ((long)z * a.Npad + q) * a.ncols + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 u = *reinterpret_cast<const float4*>(a.U + (long)q * a.ldu + c);
    float4 g;
    g.x = acc.x * inv_l * u.x;
    g.y = acc.y * inv_l * u.y;
    g.z = acc.z * inv_l * u.z;
    g.w = acc.w * inv_l * u.w;
    *reinterpret_cast<float4*>(a.G + (long)q * a.ldg + c) = g;
  }
}

__global__ __launch_bounds__(256) void combine_kernel(rmem_combine_args a) {
  __shared__ float slot_sum[64];
  combine_body(a, blockIdx.x, slot_sum);
}

__global__ __launch_bounds__(256) void combine2_kernel(rmem_combine_args a, rmem_combine_args b) {
  __shared__ float slot_sum[64];
  if ((int)blockIdx.x < a.N) combine_body(a, blockIdx.x, slot_sum);
  else combine_body(b, blockIdx.x - a.N, slot_sum);
}

static int combine_args_ok(const rmem_combine_args& a) {
  if (a.N <= 0 || a.T <= 0 || a.T > 64 || (a.ncols % 4) != 0 || !a.part || !a.lpart || !a.U || !a.G) return 0;
  if ((a.ldu % 4) || (a.ldg % 4)) return 0;
  return 1;
}

extern "C" int rmem_attn_combine2(const rmem_combine_args* ap, const rmem_combine_args* bp, void* stream) {
  if (!ap || !bp || !combine_args_ok(*ap) || !combine_args_ok(*bp)) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(combine2_kernel, dim3(ap->N + bp->N), dim3(256), 0, static_cast<hipStream_t>(stream), *ap, *bp);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_attn_combine(const rmem_combine_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  const rmem_combine_args& a = *ap;
  if (a.N <= 0 || a.T <= 0 || a.T > 64 || (a.ncols % 4) != 0 || !a.part || !a.lpart || !a.U || !a.G)
    return RMEM_ERR_INVALID;
  if ((a.ldu % 4) || (a.ldg % 4)) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(combine_kernel, dim3(a.N), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ temporal-PE bias
struct PeRows { int row[16]; };

struct PeBiasArgs {
  const float* Q; long ldq; const float* cur_pe; const float* mem_pe; PeRows rows; int T, N, d; float* bias;
};
__device__ void pe_bias_kernel(const PeBiasArgs& a, int) {
  const float* Q = a.Q; const long ldq = a.ldq; const float* cur_pe = a.cur_pe; const float* mem_pe = a.mem_pe;
  const PeRows& rows = a.rows; const int T = a.T, N = a.N, d = a.d; float* bias = a.bias;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= N) return;
  for (int t = 0; t < T; ++t) {
    float s = 0.f;
    for (int c = lane; c < d; c += 64)
      s += (Q[(long)q * ldq + c] + cur_pe[c]) * mem_pe[(long)rows.row[t] * d + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) bias[(long)q * T + t] = s;
  }
}

extern "C" int rmem_pe_bias(const float* Q, int64_t ldq, const float* cur_pe, const float* mem_pe,
                            const int32_t* pe_row_host, int32_t T, int32_t N, int32_t d, float* bias,
                            void* stream) {
  if (!Q || !cur_pe || !mem_pe || !pe_row_host || !bias || T <= 0 || T > 16 || N <= 0) return RMEM_ERR_INVALID;
  PeBiasArgs a{Q, (long)ldq, cur_pe, mem_pe, {}, T, N, d, bias};
  for (int t = 0; t < 16; ++t) a.rows.row[t] = t < T ? pe_row_host[t] : 0;
  return rmem::launch<PeBiasArgs, pe_bias_kernel, 256>(a, dim3((N + 3) / 4), dim3(256), 0,
                                                        static_cast<hipStream_t>(stream));
}
