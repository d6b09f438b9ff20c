// fused.hip -- the memory read of GatedPropagation / LocalGatedPropagation (layers/attention.py:
// 174-209, 289-358) as ONE flash-style launch per read: S = Q.K^T, online softmax and O = P.V
// without ever writing the probability matrix to HBM (the three-launch form of attn.hip writes and
// re-reads 25-50 MB of P per read and computes Q.K^T twice).
//
// Work unit = (128-query tile, 512-column half of [V | ID_V], key split).  One workgroup = 4 waves,
// ONE wave per SIMD with the whole 512-register file: the wave's O accumulator is 128 queries x 128
// columns = 4 x 4 MFMA tiles = 256 accumulator registers, i.e. the unit's accumulators fill half of
// the CU's register file -- the largest output tile the chip can hold, which minimises the bytes
// of V streamed per MFMA (every V element is used by 128 queries x 3 products).
//
// Per 64-key step:
//   (1) S^T[key][q] = K.Q^T : wave w owns query sub-tile w (32 queries) x 64 keys; K tile and Q
//       tile come from LDS (XOR-swizzled 128-byte rows, the layout of gemm_core.h); "swapped"
//       (rows = keys) so that the softmax statistics of a query live in one lane (+ lane^32);
//   (2) online softmax in fp32: running max m, sum l, per-slot sums (attention mass); P = exp(S-m)
//       goes to LDS as fp16 hi/lo planes [128 q][64 keys]; the rescale factors exp(m_old - m_new)
//       of the 128 queries go to LDS with a per-sub-tile "changed" flag;
//   (3) barrier; O[q][c] += P.V : A = P fragments from LDS (shared by the four waves), B = V
//       fragments straight from global memory into registers: V is stored "blocked-16"
//       [slot][key/16][col][16 keys], so the B fragment of a 32-column tile (lane = column, 8
//       consecutive keys) is ONE contiguous KiB per load instruction, and a wave's 128 columns x 16
//       keys are 4 contiguous KiB -- no LDS round trip for the operand that is not shared between
//       waves; the next k-step's fragments are in flight while this one is multiplied.
// Split precision: every product is hi*lo' + lo*hi' + hi*hi' on v_mfma_f32_32x32x16_f16.
// Key splits write un-normalised partials + (max, sum) [+ per-slot (sum, max)]; rmem_attn_read_combine
// merges them, normalises, gates with U and emits the per-slot attention mass.
#include "../../include/rmem_hip.h"
#include "gemm_core.h"
#include "attn_common.h"
#include "launch.h"

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

constexpr int RD_Q_BYTES = 4 * 128 * 128;   // [plane][k half][128 queries][128 B]
constexpr int RD_K_BYTES = 4 * 64 * 128;    // [plane][k half][64 keys][128 B]
constexpr int RD_P_BYTES = 2 * 128 * 128;   // [plane][128 queries][64 keys fp16]
constexpr int RD_FAC_OFF = RD_Q_BYTES + RD_K_BYTES + RD_P_BYTES;
constexpr int RD_LDS = RD_FAC_OFF + 128 * 4 + 16;
constexpr float RD_NEG = -3.0e38f;
constexpr float RD_THR = 14.0f;               // log2 domain: weights up to 2^14 = 16384 < 65504 (fp16 hi plane)

#define RD_OPAQUE(x) asm volatile("" : "+v"(x))

template <int TRACE>
__device__ __forceinline__ void read_body(const rmem_read_args& a, const int blk, char* smem) {
  const int MODE = a.mode;                            // wave-uniform: one code path serves both reads
  constexpr int KS_OFF = RD_Q_BYTES, PS_OFF = RD_Q_BYTES + RD_K_BYTES;
  float* fac = reinterpret_cast<float*>(smem + RD_FAC_OFF);
  int* flag = reinterpret_cast<int*>(smem + RD_FAC_OFF + 512);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, j = lane & 31;

  // ---- work unit.  Units are ordered (split, column half, query tile) and every XCD (block b runs
  // on XCD b % 8: observed placement, used for speed only) owns a contiguous chunk, so the units of
  // one XCD share a key split and a column half, i.e. the same V bytes in that XCD's L2.
  const int nq = a.Npad / 128, ncp = a.ncols / 512;
  const int nunits = nq * ncp * a.ksplits;
  const int chunk = (nunits + 7) / 8;
  const int jj = blk >> 3;
  const int u = (blk & 7) * chunk + jj;
  if (jj >= chunk || u >= nunits) return;
  const int qtile = u % nq;
  const int cp = (u / nq) % ncp;
  const int z = u / (nq * ncp);

  const int tv = (a.N + 63) / 64;            // 64-key tiles of a slot that hold valid keys
  int k_lo, k_hi;
  if (MODE == 0) {
    k_lo = 0;
    k_hi = a.T * tv;
  } else {
    int t_lo, t_hi;
    band_tiles(qtile, a.N, a.h, a.w, t_lo, t_hi);
    k_lo = 2 * t_lo;
    k_hi = 2 * t_hi < tv ? 2 * t_hi : tv;
  }
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  const int lo = k_lo + z * per;
  const int hi_t = lo + per < k_hi ? lo + per : k_hi;

  const int q = qtile * 128 + wave * 32 + j;          // this lane's query in the score phase
  const bool qvalid = q < a.N;
  const bool writer = cp == 0 && hi == 0;             // statistics are written once per (split, query)
  float* mlp = a.ml + ((long)z * a.Npad + q) * 2;
  // debug aid (read_kernel<1> only): a tracing launch receives 16 shader-clock stamps per block through R (bank
  // mode, where R is unused) or through lslot (window mode, which never records per-slot sums in the product)
  const bool tracing = TRACE && ((MODE == 0 && a.R) || (MODE == 1 && a.lslot));
  float* lsp = (a.lslot && !(TRACE && MODE == 1)) ? a.lslot + ((long)z * a.Npad + q) * a.T * 2 : nullptr;
  if (lo >= hi_t) {                                   // no key tile in this split (narrow band)
    if (writer) {
      mlp[0] = RD_NEG;
      mlp[1] = 0.f;
    }
    return;
  }
  if (lsp && writer)
    for (int t = 0; t < a.T; ++t) {
      lsp[2 * t] = 0.f;
      lsp[2 * t + 1] = 0.f;
    }

  long long* trace = tracing ? reinterpret_cast<long long*>(MODE == 0 ? const_cast<float*>(a.R) : a.lslot) + (long)blk * 16 : nullptr;
  long long tph[5] = {0, 0, 0, 0, 0}, tlast = 0;      // TRACE: cycles in score / softmax / barrier A / P.V / barrier B
#define RD_STAMP(k) do { if (TRACE) { const long long t__ = __builtin_readcyclecounter(); tph[k] += t__ - tlast; tlast = t__; } } while (0)
  if (trace && tid == 0) trace[0] = __builtin_readcyclecounter();

  SlotLut lut;
  lut.load(a.slot_map, MODE == 0 ? a.T : 1);

  // ---- LDS addresses.  Every image has 128-byte rows with 16-byte chunk c of row r stored at
  // c ^ ((r >> 1) & 7) (gemm_core.h).  All rows a lane touches are  j (mod 32), so the swizzle term
  // depends on the lane only and each family of reads is ONE address register per k-step plus
  // immediate offsets (computed once and kept opaque: left to itself the compiler hoists ~150
  // per-read addresses out of the loop and spills them).
  const int sw = (j >> 1) & 7;
  int af[4];                                          // fragment chunk (2k + hi) of row j
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    af[k] = j * 128 + (((2 * k + hi) ^ sw) << 4);
    RD_OPAQUE(af[k]);
  }
  int aq[4], ak[4], ap[4];                            // Q rows of this wave / K rows / P rows
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    aq[k] = af[k] + wave * 4096;
    ak[k] = af[k] + KS_OFF;
    ap[k] = af[k] + PS_OFF;
    RD_OPAQUE(aq[k]);
    RD_OPAQUE(ak[k]);
    RD_OPAQUE(ap[k]);
  }
  int apst[8];                                        // P stores: row wave*32 + j, chunk c = 0..7, half hi
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    apst[c] = PS_OFF + wave * 4096 + j * 128 + ((c ^ sw) << 4) + hi * 8;
    RD_OPAQUE(apst[c]);
  }
  // staging stores: thread -> (row tid >> 4 (+16 per pass), chunk tid & 15 of the 256-byte row)
  const int st_sw = (tid >> 5) & 7;
  int ast = ((tid >> 3) & 1) * 0 + (tid >> 4) * 128 + (((tid & 7) ^ st_sw) << 4);
  RD_OPAQUE(ast);
  const int st_half = (tid >> 3) & 1;                 // k half of this thread's chunk
  int goff = (tid >> 4) * 128 + (tid & 15) * 8;       // element offset inside a [rows][128] plane
  RD_OPAQUE(goff);

  // ---- Q tile -> LDS (once): 2 planes x 128 rows x 16 chunks, 8 passes of 16 rows per plane
  {
    const h16_t* qbase[2] = {a.qh + (long)qtile * 128 * 128, a.ql + (long)qtile * 128 * 128};
    static_for<16>([&](auto P) {
      constexpr int p = P.value;
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(qbase[p >> 3] + (p & 7) * 16 * 128 + goff);
      *reinterpret_cast<u32x4_t*>(smem + ast + st_half * 16384 + ((p >> 3) * 32768 + (p & 7) * 2048)) = v;
    });
  }

  if (trace && tid == 0) trace[11] = __builtin_readcyclecounter();   // Q tile requested and stored
  // ---- staging helpers
  u32x4_t kr[8];
  auto tile_slot = [&](int i, int& key0) __attribute__((always_inline)) {
    int t = 0, kt = i;
    if (MODE == 0) {
      t = i / tv;
      kt = i - t * tv;
    }
    key0 = kt * 64;
    return t;
  };
  auto kload = [&](int i) __attribute__((always_inline)) {
    int key0;
    const int t = tile_slot(i, key0);
    const long base = (long)lut(t) * a.k_slot_stride + (long)key0 * 128;
    const h16_t* b0 = a.kh + base;
    const h16_t* b1 = a.kl + base;
    static_for<8>([&](auto P) {
      constexpr int p = P.value;
      kr[p] = *reinterpret_cast<const u32x4_t*>(((p >> 2) ? b1 : b0) + (p & 3) * 16 * 128 + goff);
    });
  };
  auto kstore = [&]() __attribute__((always_inline)) {
    static_for<8>([&](auto P) {
      constexpr int p = P.value;
      *reinterpret_cast<u32x4_t*>(smem + ast + st_half * 8192 + (KS_OFF + (p >> 2) * 16384 + (p & 3) * 2048)) = kr[p];
    });
  };
  // V fragments of k-step ks of tile i: [ci][plane], lane = column, 8 consecutive keys (16 B)
  u32x4_t va[8], vb[8];
  int vcol = (wave * 128 + j) * 16 + hi * 8;
  RD_OPAQUE(vcol);
  auto vload = [&](u32x4_t (&dst)[8], int i, int ks) __attribute__((always_inline)) {
    int key0;
    const int t = tile_slot(i, key0);
    const long base = (long)lut(t) * a.v_slot_stride + ((long)((key0 >> 4) + ks) * a.ncols + cp * 512) * 16;
    const h16_t* b0 = a.vh + base;
    const h16_t* b1 = a.vl + base;
    static_for<4>([&](auto CI) {
      constexpr int ci = CI.value;
      dst[ci * 2 + 0] = *reinterpret_cast<const u32x4_t*>(b0 + ci * 32 * 16 + vcol);
      dst[ci * 2 + 1] = *reinterpret_cast<const u32x4_t*>(b1 + ci * 32 * 16 + vcol);
    });
  };

  // m (and the stored (max, sum) statistics) live in the log2 domain until they are written
  const float sl2e = a.scale * 1.44269504088896341f;
  constexpr float LN2 = 0.693147180559945f;
  float m = RD_NEG, l = 0.f, lcur = 0.f, bias_t = 0.f;
  int cur_t = -1;
  int qy = 0, qx = 0;
  const float inv_w = MODE == 1 ? 1.0f / (float)a.w : 0.f;   // (a.w >= 1 is validated for mode 1)
  const float* Rq = nullptr;
  const int rcs = a.rcs > 0 ? a.rcs : 1;
  if (MODE == 1) {
    qy = fast_div(qvalid ? q : 0, inv_w);
    qx = (qvalid ? q : 0) - qy * a.w;
    Rq = a.R + (long)(qvalid ? q : 0) * a.ldr;
  }

  // Windowed read: relative bias (log2 domain) and visibility of this lane's 32 keys of the tile at
  // key0.  All 32 gathers are issued unconditionally (index 0 where masked) and back to back: written
  // per element under a mode / validity branch they complete one memory latency after the other
  // (measured: a windowed tile took 28 us, four long-term tiles).  One division per 4 consecutive keys.
  auto window_terms = [&](int key0, float (&rb)[32]) __attribute__((always_inline)) {
    int idx[32];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int tok0 = key0 + sub * 32 + 8 * g + 4 * hi;
        const int ky0 = fast_div(tok0, inv_w);
        const int kx0 = tok0 - ky0 * a.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ky = ky0, kx = kx0 + e;
          if (kx >= a.w) {
            kx -= a.w;
            ky += 1;
          }
          const int dy = ky - qy, dx = kx - qx;
          const bool valid = qvalid && tok0 + e < a.N && dy >= -7 && dy <= 7 && dx >= -7 && dx <= 7;
          idx[sub * 16 + g * 4 + e] = valid ? ((dy + 7) * 15 + dx + 7) * rcs : -1;
        }
      }
#pragma unroll
    for (int r = 0; r < 32; ++r) rb[r] = Rq[idx[r] < 0 ? 0 : idx[r]];
#pragma unroll
    for (int r = 0; r < 32; ++r) rb[r] = idx[r] < 0 ? RD_NEG : rb[r] * 1.44269504088896341f;   // RD_NEG marks "masked"
  };

  // ---- reference pass: m = (approximate) row maximum of the scores over the whole split, from the hi
  // planes only (one product instead of three; |error| ~ 2^-10 of |q||k| scale, a few hundredths;
  // the relative bias of the windowed read is added exactly: a reference ABOVE the maximum by D bits
  // pushes the lo plane of every weight D bits into the fp16 subnormals -- 1.7e-4 on the LSTT output
  // when the bias was simply dropped -- and one far below it ends the segment early).
  // The weights of the main pass are exp2(y - m) against this FIXED reference, so they stay within a
  // few per cent of 1 at the row maximum and the accumulators never need a rescale.  (A reference
  // taken from the first tile alone made the deferred-rescale path below data dependent: peaked
  // attention, whose maximum sits in a late tile, broke every unit into 2-3 segments -- 2.3x slower
  // in the engine than on random data.)  Costs 16 MFMAs + one hi-plane K staging per tile.
  {
    float mest = RD_NEG;
    int pt = -1;
    float pb2 = 0.f;
    // hi plane only, two tiles resident in the K region of the LDS (buffer b at +16 KB * b): the next
    // tile is requested before this one is multiplied and stored after it -- one barrier per tile
    u32x4_t hr[4];
    auto hload = [&](int i) __attribute__((always_inline)) {
      int key0;
      const int t = tile_slot(i, key0);
      const h16_t* b0 = a.kh + (long)lut(t) * a.k_slot_stride + (long)key0 * 128;
      static_for<4>([&](auto P) { hr[P.value] = *reinterpret_cast<const u32x4_t*>(b0 + P.value * 16 * 128 + goff); });
    };
    auto hstore = [&](int b) __attribute__((always_inline)) {
      static_for<4>([&](auto P) {
        *reinterpret_cast<u32x4_t*>(smem + ast + st_half * 8192 + b * 16384 + (KS_OFF + P.value * 2048)) = hr[P.value];
      });
    };
    hload(lo);
    hstore(0);
    __syncthreads();
    for (int i = lo; i < hi_t; ++i) {
      int key0;
      const int t = tile_slot(i, key0);
      const int boff = ((i - lo) & 1) * 16384;
      if (i + 1 < hi_t) hload(i + 1);
      if (MODE == 0 && t != pt) {
        pt = t;
        pb2 = ((a.bias && qvalid) ? a.bias[(long)q * a.T + t] : 0.f) * sl2e;
      }
      f32x16_t s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
      frag8_t g0[3], g1[3];                           // Q hi, K hi (sub 0), K hi (sub 1)
      auto gload = [&](frag8_t (&g)[3], auto KS) __attribute__((always_inline)) {
        constexpr int ks = KS.value;
        constexpr int kh = ks >> 2, k4 = ks & 3;
        g[0] = *reinterpret_cast<const frag8_t*>(smem + aq[k4] + kh * 16384);
        g[1] = *reinterpret_cast<const frag8_t*>(smem + ak[k4] + boff + kh * 8192);
        g[2] = *reinterpret_cast<const frag8_t*>(smem + ak[k4] + boff + (kh * 8192 + 4096));
      };
      gload(g0, std::integral_constant<int, 0>{});
      static_for<8>([&](auto KS) {
        constexpr int ks = KS.value;
        frag8_t (&cur)[3] = (ks & 1) ? g1 : g0;
        frag8_t (&nxt)[3] = (ks & 1) ? g0 : g1;
        if constexpr (ks < 7) gload(nxt, std::integral_constant<int, (ks < 7 ? ks + 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        s0 = RMEM_MFMA(cur[1], cur[0], s0);
        s1 = RMEM_MFMA(cur[2], cur[0], s1);
        __builtin_amdgcn_sched_barrier(0);
      });
      const bool padded = key0 + 64 > a.N;
      if (MODE == 0 && !padded) {
        float tm = s0[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, fmaxf(s0[r], s1[r]));
        mest = fmaxf(mest, fmaf(tm, sl2e, pb2));
      } else if (MODE == 0) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tok = key0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float sv = fmaf(sub ? s1[r] : s0[r], sl2e, pb2);
            mest = fmaxf(mest, tok < a.N ? sv : RD_NEG);
          }
      } else {
        float rb[32];
        window_terms(key0, rb);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float rbv = rb[sub * 16 + r];
            const float sv = fmaf(sub ? s1[r] : s0[r], sl2e, rbv);
            mest = fmaxf(mest, rbv > -2.9e38f ? sv : RD_NEG);
          }
      }
      if (i + 1 < hi_t) hstore(((i - lo) + 1) & 1);    // the other buffer: its readers passed the last barrier
      __syncthreads();
    }
    mest = fmaxf(mest, __shfl_xor(mest, 32));
    if (mest > -2.9e38f) m = mest;
  }
  if (trace && tid == 0) {
    trace[9] = __builtin_readcyclecounter();           // end of the reference pass
    trace[10] = hi_t - lo;                             // key tiles of this unit
  }
  kload(lo);
  kstore();
  __syncthreads();

  // Online softmax with the accumulator rescale DEFERRED THROUGH MEMORY.  The unit's accumulators
  // fill the accumulator file exactly, and this compiler cannot update them in place under a branch
  // (it copies the tuples and spills 240 registers), so O is never rescaled in registers.  A
  // "segment" is a run of key tiles accumulated against one fixed per-row reference m (weights
  // exp(s - m) <= e^RD_THR: fp16 hi/lo planes carry them at full relative precision).  When a tile's
  // row maximum exceeds m + RD_THR the segment ends BEFORE that tile: its O is flushed to the
  // partial buffer (first segment: plain store = the normal epilogue; later ones: part = part *
  // exp(m_prev - m) + O, rows owned by this wave only) and a new segment starts at that tile with
  // m = max(m, tile max).  With one segment per unit -- the usual case -- nothing extra runs.
  if (trace && tid == 0) trace[1] = __builtin_readcyclecounter();
  int i = lo;
  bool first_seg = true;
  float* out = a.part + ((long)z * a.Npad + qtile * 128) * a.ncols + (long)cp * 512 + wave * 128 + j;
  while (i < hi_t) {
  f32x16_t o[4][4];
#pragma unroll
  for (int qi = 0; qi < 4; ++qi)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qi][ci][r] = 0.f;
  const int i0 = i;
  for (; i < hi_t; ++i) {
    int key0;
    const int t = tile_slot(i, key0);
    if (TRACE) tlast = __builtin_readcyclecounter();
    vload(va, i, 0);                                  // in flight during the score phase
    if (i + 1 < hi_t) kload(i + 1);
    if (MODE == 0 && t != cur_t) {
      if (lsp && writer && cur_t >= 0) {
        lsp[2 * cur_t] = lcur;
        lsp[2 * cur_t + 1] = m * LN2;
      }
      lcur = 0.f;
      cur_t = t;
      bias_t = (a.bias && qvalid) ? a.bias[(long)q * a.T + t] : 0.f;   // (padding queries: no row in bias)
    }
    // ---- (1) S^T = K . Q^T for this wave's 32 queries x 64 keys.  Fragments are requested two d-steps
    // ahead of their MFMAs (three rotating register sets); the scheduling barriers keep the compiler
    // from hoisting all 48 fragment reads to the top (192 registers -> spills).
    f32x16_t s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
    {
      frag8_t f0[6], f1[6], f2[6];                    // [0,1] = Q hi/lo, [2..5] = K (sub, plane)
      auto fload = [&](frag8_t (&ld)[6], auto KS) __attribute__((always_inline)) {
        constexpr int ks = KS.value;
        constexpr int kh = ks >> 2, k4 = ks & 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          ld[p] = *reinterpret_cast<const frag8_t*>(smem + aq[k4] + (p * 2 + kh) * 16384);
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
            ld[2 + sub * 2 + p] = *reinterpret_cast<const frag8_t*>(smem + ak[k4] + ((p * 2 + kh) * 8192 + sub * 4096));
        }
      };
      fload(f0, std::integral_constant<int, 0>{});
      fload(f1, std::integral_constant<int, 1>{});
      static_for<8>([&](auto KS) {
        constexpr int ks = KS.value;
        frag8_t (&cur)[6] = (ks % 3 == 0) ? f0 : (ks % 3 == 1) ? f1 : f2;
        frag8_t (&ld)[6] = ((ks + 2) % 3 == 0) ? f0 : ((ks + 2) % 3 == 1) ? f1 : f2;
        if constexpr (ks + 2 < 8) fload(ld, std::integral_constant<int, (ks + 2 < 8 ? ks + 2 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        s[0] = RMEM_MFMA(cur[2], cur[1], s[0]);
        s[1] = RMEM_MFMA(cur[4], cur[1], s[1]);
        s[0] = RMEM_MFMA(cur[3], cur[0], s[0]);
        s[1] = RMEM_MFMA(cur[5], cur[0], s[1]);
        s[0] = RMEM_MFMA(cur[2], cur[0], s[0]);
        s[1] = RMEM_MFMA(cur[4], cur[0], s[1]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    RD_STAMP(0);
    // ---- (2) online softmax of this lane's query (keys of lane^32 are the other half of the tile), in
    // the log2 domain: y = (s + bias) * scale * log2(e), weights 2^(y - m).  Token masks only on the
    // tile of a slot that holds padding (wave-uniform); rows of padding QUERIES are computed like any
    // other (finite garbage that nobody reads).
    const bool padded = key0 + 64 > a.N;
    float tmax = RD_NEG;
    if (MODE == 0 && !padded) {
      const float b2 = bias_t * sl2e;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sv = fmaf(s[sub][r], sl2e, b2);
          s[sub][r] = sv;
          tmax = fmaxf(tmax, sv);
        }
    } else if (MODE == 0) {
      const float b2 = bias_t * sl2e;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tok = key0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float sv = tok < a.N ? fmaf(s[sub][r], sl2e, b2) : RD_NEG;
          s[sub][r] = sv;
          tmax = fmaxf(tmax, sv);
        }
    } else {
      float rb[32];
      window_terms(key0, rb);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float rbv = rb[sub * 16 + r];
          const float sv = rbv > -2.9e38f ? fmaf(s[sub][r], sl2e, rbv) : RD_NEG;
          s[sub][r] = sv;
          tmax = fmaxf(tmax, sv);
        }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    float alpha = 1.0f;
    bool need_break = false;
    if (i == i0) {                                    // first tile of a segment: adopt the new reference
      const float m_new = fmaxf(m, tmax);
      alpha = __builtin_amdgcn_exp2f(m - m_new);      // 0 on the very first valid tile
      m = m_new;
      if (hi == 0) fac[wave * 32 + j] = alpha;        // factor of the flushed partials, used at the flush
    } else if (m < -2.9e38f) {
      m = tmax;                                       // no valid key so far: O row and l are still zero
    } else {
      need_break = tmax > m + RD_THR;
    }
    float psum = 0.f;
    // weights of 4 consecutive keys -> fp16 hi / lo planes of the P image
    auto emit4 = [&](const f32x2_t (&pp)[2], int sub, int g) __attribute__((always_inline)) {
      // hi = fp16(p), lo = fp16(p - hi): packed conversions (v_cvt_pk_f16_f32, round to nearest even)
      const f16x2_t h0 = __builtin_convertvector(pp[0], f16x2_t), h1 = __builtin_convertvector(pp[1], f16x2_t);
      const f32x2_t r0 = pp[0] - __builtin_convertvector(h0, f32x2_t), r1 = pp[1] - __builtin_convertvector(h1, f32x2_t);
      const f16x2_t l0 = __builtin_convertvector(r0, f16x2_t), l1 = __builtin_convertvector(r1, f16x2_t);
      u32x2_t wh, wl;
      wh[0] = __builtin_bit_cast(uint32_t, h0);
      wh[1] = __builtin_bit_cast(uint32_t, h1);
      wl[0] = __builtin_bit_cast(uint32_t, l0);
      wl[1] = __builtin_bit_cast(uint32_t, l1);
      // keys sub*32 + 8g + 4hi + 0..3 of row wave*32 + j: chunk sub*4 + g, half hi
      *reinterpret_cast<u32x2_t*>(smem + apst[sub * 4 + g]) = wh;
      *reinterpret_cast<u32x2_t*>(smem + apst[sub * 4 + g] + 16384) = wl;
    };
    // The masked form (windowed read, tiles with padding keys) lives on its own branch: as a select
    // under a wave-uniform condition it is if-converted into a compare and two v_cndmask per element
    // on EVERY tile (96 of the ~400 VALU instructions of this phase, found in the ISA).
    if (!(MODE == 1 || padded)) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x2_t pp[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __builtin_amdgcn_exp2f(s[sub][4 * g + e] - m);
            psum += p;
            pp[e >> 1][e & 1] = p;
          }
          emit4(pp, sub, g);
        }
    } else {
      asm volatile("" ::: "memory");                  // keeps this block a branch
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x2_t pp[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sv = s[sub][4 * g + e];
            float p = __builtin_amdgcn_exp2f(sv - m);   // sentinels (-3e38) give exactly 0 unless m is one too
            p = sv > -2.9e38f ? p : 0.f;
            psum += p;
            pp[e >> 1][e & 1] = p;
          }
          emit4(pp, sub, g);
        }
    }
    psum += __shfl_xor(psum, 32);
    const int any_break = __any(need_break) ? 1 : 0;
    if (lane == 0) flag[wave] = any_break;
    RD_STAMP(1);
    __syncthreads();                                  // P, flags visible; everyone is done with the K tile
    RD_STAMP(2);
    if (__builtin_amdgcn_readfirstlane(flag[0] | flag[1] | flag[2] | flag[3])) {
      __syncthreads();                                // flags are rewritten by the redone tile
      break;                                          // the K tile in LDS is kept: tile i is redone
    }
    l = l * alpha + psum;
    lcur = lcur * alpha + psum;
    if (i + 1 < hi_t) kstore();
    // ---- (3) O += P . V for this wave's 128 columns, all 128 queries
    {
      frag8_t pa[8], pb[8];                           // P fragments [qi][plane] of one k-step
      static_for<5>([&](auto KS) {
        constexpr int ks = KS.value;                  // iteration ks: request step ks, multiply step ks-1
        frag8_t (&pld)[8] = (ks & 1) ? pb : pa;
        frag8_t (&pc)[8] = (ks & 1) ? pa : pb;
        u32x4_t (&vld)[8] = (ks & 1) ? vb : va;       // step 0 was requested before the score phase
        u32x4_t (&vc)[8] = (ks & 1) ? va : vb;
        if constexpr (ks < 4) {
          if constexpr (ks > 0) vload(vld, i, ks);
#pragma unroll
          for (int qi = 0; qi < 4; ++qi)
#pragma unroll
            for (int p = 0; p < 2; ++p)
              pld[qi * 2 + p] = *reinterpret_cast<const frag8_t*>(smem + ap[ks] + (p * 16384 + qi * 4096));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks > 0) {
#pragma unroll
          for (int qi = 0; qi < 4; ++qi)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {   // small terms first
              const frag8_t vh = __builtin_bit_cast(frag8_t, vc[ci * 2 + 0]);
              const frag8_t vl = __builtin_bit_cast(frag8_t, vc[ci * 2 + 1]);
              o[qi][ci] = RMEM_MFMA(pc[qi * 2 + 0], vl, o[qi][ci]);
              o[qi][ci] = RMEM_MFMA(pc[qi * 2 + 1], vh, o[qi][ci]);
              o[qi][ci] = RMEM_MFMA(pc[qi * 2 + 0], vh, o[qi][ci]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // keep the accumulators in the accumulator half of the register file (the flush below is VALU
    // work: left alone the compiler re-classes them as arch VGPRs and shuffles / spills hundreds)
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) asm volatile("" : "+a"(o[qi][ci]));
    RD_STAMP(3);
    __syncthreads();                                  // P / flags may be overwritten; next K tile visible
    RD_STAMP(4);
  }
  if (trace && tid == 0 && first_seg) trace[2] = __builtin_readcyclecounter();
  // ---- flush the segment [i0, i)
  if (first_seg && i >= hi_t) {
    // The usual case: one segment.  Transposed through LDS (the Q / K / P images are dead: every wave is past the last barrier of
    // the tile loop): the accumulator layout gives one column per lane, i.e. 4-byte stores; staged
    // as [64 rows][128 columns] per wave, two rounds, the tile leaves as 16-byte stores of whole
    // 512-byte rows (the 4-byte form was store-issue-bound: ~25 us of a 113 us launch).
    char* stg = smem + wave * 32768;
    int lrow = hi, lcol = j;                          // opaque here: keeps this block's 64 row pointers out of the prologue
    RD_OPAQUE(lrow);
    RD_OPAQUE(lcol);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = q2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
#pragma unroll
          for (int ci = 0; ci < 4; ++ci)
            *reinterpret_cast<float*>(stg + row * 512 + (ci * 32 + lcol) * 4) = o[half * 2 + q2][ci][r];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave: LDS operations complete in order
      float* orow = a.part + ((long)z * a.Npad + qtile * 128 + half * 64) * a.ncols + (long)cp * 512 + wave * 128;
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const int row = it * 2 + lrow;
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(stg + row * 512 + lcol * 16);
        *reinterpret_cast<f32x4_t*>(orow + (long)row * a.ncols + lcol * 4) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next round overwrites
    }
  } else {
    // (the lane term is made opaque HERE: otherwise the compiler computes the 64 factor addresses and
    // the 256 row offsets of this rare block before the tile loop and spills them -- 237 registers x
    // 65536 threads = 62 MB of scratch writes per launch, measured as WRITE_SIZE 162 MB vs 66 MB of partials)
    int hi4 = 4 * hi;
    RD_OPAQUE(hi4);
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qi * 32 + (r & 3) + 8 * (r >> 2) + hi4;
        const float f = fac[qq];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          float* pp = out + (long)qq * a.ncols + ci * 32;
          float ov;   // read through asm: keeps the AGPR -> VGPR copies inside this rare block
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ov) : "a"(o[qi][ci][r]));
          *pp = (first_seg ? 0.f : *pp * f) + ov;     // the first flush initialises the partial
        }
      }
    __syncthreads();                                  // fac is rewritten by the next segment's first tile
  }
  first_seg = false;
  }

  if (trace && tid == 0) {
    trace[3] = __builtin_readcyclecounter();
    for (int k = 0; k < 5; ++k) trace[4 + k] = tph[k];
  }
  // ---- statistics
  if (lsp && writer && cur_t >= 0) {
    lsp[2 * cur_t] = lcur;
    lsp[2 * cur_t + 1] = m * LN2;
  }
  if (writer) {
    mlp[0] = m > -2.9e38f ? m * LN2 : RD_NEG;
    mlp[1] = l;
  }
}

template <int TRACE>
__global__ __launch_bounds__(256, 1) void read_kernel(rmem_read_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  read_body<TRACE>(a, blockIdx.x, smem);
}

// The bank read (p[0], mode 0) and the windowed read (p[1], mode 1) of one layer in ONE launch.  Per
// XCD (block % 8) the first cha blocks serve p[0]'s units, the next chb blocks p[1]'s.
struct Read2Args {
  rmem_read_args p[2];
  int cha, chb;
};

__global__ __launch_bounds__(256, 1) void read2_kernel(Read2Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int which = jj < g.cha ? 0 : 1;
  read_body<0>(g.p[which], (which ? jj - g.cha : jj) * 8 + xcd, smem);
}

// the same two kernels for several clips in one launch (launch.h): block z = clip, whose argument
// block is read from device memory
__global__ __launch_bounds__(256, 1) void read_many_kernel(const char* __restrict__ argv, long stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const rmem_read_args a = rmem::uniform_copy(reinterpret_cast<const rmem_read_args*>(argv + (long)blockIdx.z * stride));
  read_body<0>(a, blockIdx.x, smem);
}

__global__ __launch_bounds__(256, 1) void read2_many_kernel(const char* __restrict__ argv, long stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Read2Args& g = *reinterpret_cast<const Read2Args*>(argv + (long)blockIdx.z * stride);
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int which = jj < __builtin_amdgcn_readfirstlane(g.cha) ? 0 : 1;
  const rmem_read_args a = rmem::uniform_copy(&g.p[which]);
  read_body<0>(a, (which ? jj - g.cha : jj) * 8 + xcd, smem);
}

template <class K>
static int read_many_thunk(K kernel, const rmem::RecOp& op, const char* dev_args, long stride, int B, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS);
  hipLaunchKernelGGL(kernel, dim3(op.grid.x, 1, B), dim3(256), RD_LDS, s, dev_args + op.off, stride);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}
static int read_many(const rmem::RecOp& op, const char* d, long st, int B, hipStream_t s) {
  return read_many_thunk(&read_many_kernel, op, d, st, B, s);
}
static int read2_many(const rmem::RecOp& op, const char* d, long st, int B, hipStream_t s) {
  return read_many_thunk(&read2_many_kernel, op, d, st, B, s);
}

static int read_args_ok(const rmem_read_args& a) {
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) != 0 || a.T <= 0 || a.T > 16 || a.ksplits <= 0 || a.ksplits > 32) return 0;
  if (!a.qh || !a.ql || !a.kh || !a.kl || !a.vh || !a.vl || !a.part || !a.ml) return 0;
  if (a.ncols <= 0 || (a.ncols % 512) != 0) return 0;
  if (a.mode == 1 && (!a.R || a.h * a.w != a.N || a.T != 1 || a.w < 1 || a.ldr < 1)) return 0;
  if (a.mode != 0 && a.mode != 1) return 0;
  return 1;
}

static int read_chunk(const rmem_read_args& a) {
  return ((a.Npad / 128) * (a.ncols / 512) * a.ksplits + 7) / 8;
}

extern "C" int rmem_attn_read2_v128(const rmem_read_args* ap, const rmem_read_args* bp, void* stream) {
  if (!ap || !bp || !read_args_ok(*ap) || !read_args_ok(*bp) || ap->mode != 0 || bp->mode != 1) return RMEM_ERR_INVALID;
  const int cha = read_chunk(*ap), chb = read_chunk(*bp);
  // per launch: the attribute belongs to the (device, function) pair; no process-wide "already set" flag
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS);
  Read2Args g;
  g.p[0] = *ap;
  g.p[1] = *bp;
  g.cha = cha;
  g.chb = chb;
  if (rmem::Recorder* r = rmem::current_recorder()) {
    rmem::rec_push(r, &read2_many, dim3(8 * (cha + chb)), dim3(256), RD_LDS, &g, (unsigned)sizeof(g));
    return RMEM_OK;
  }
  hipLaunchKernelGGL(read2_kernel, dim3(8 * (cha + chb)), dim3(256), RD_LDS, static_cast<hipStream_t>(stream), g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_attn_read_v128(const rmem_read_args* ap, void* stream) {
  if (!ap || !read_args_ok(*ap)) return RMEM_ERR_INVALID;
  const rmem_read_args& a = *ap;
  const int chunk = read_chunk(a);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int var = ((a.mode == 0 && a.R) || (a.mode == 1 && a.lslot)) ? 1 : 0;   // the tracing build (debug aid, see read_body)
  if (rmem::Recorder* r = rmem::current_recorder()) {
    if (var) return RMEM_ERR_INVALID;
    rmem::rec_push(r, &read_many, dim3(8 * chunk), dim3(256), RD_LDS, &a, (unsigned)sizeof(a));
    return RMEM_OK;
  }
  const void* fn = var == 0 ? reinterpret_cast<const void*>(&read_kernel<0>) : reinterpret_cast<const void*>(&read_kernel<1>);
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS);
  if (var == 0) hipLaunchKernelGGL((read_kernel<0>), dim3(8 * chunk), dim3(256), RD_LDS, s, a);
  else hipLaunchKernelGGL((read_kernel<1>), dim3(8 * chunk), dim3(256), RD_LDS, s, a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ merge key splits + gate (+ mass)
// G[q][c] = U[q][c] * (sum_z w_z part[z][q][c]) / (sum_z w_z l_z),  w_z = exp(m_z - max_z m_z);
// mass[q][t] = (sum_z lslot[z][q][t].sum * exp(lslot[z][q][t].max - m)) / L   (record_attn_weight,
// layers/transformer.py:1186-1192).  Splits are visited in order: no floating-point atomics.
__device__ __forceinline__ void read_combine_body(const rmem_read_combine_args& a, int q, float* sh) {
  const int tid = threadIdx.x;
  float* wz = sh;             // [ksplits <= 32]
  float* stat = sh + 32;      // [0] = max, [1] = 1 / L
  if (tid < 64) {
    float mz = RD_NEG, lz = 0.f;
    if (tid < a.ksplits) {
      mz = a.ml[((long)tid * a.Npad + q) * 2];
      lz = a.ml[((long)tid * a.Npad + q) * 2 + 1];
      if (!(lz > 0.f)) mz = RD_NEG;
    }
    float mm = mz;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
    const float w = lz > 0.f ? expf(mz - mm) : 0.f;
    float L = w * lz;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) L += __shfl_xor(L, o);
    if (tid < a.ksplits) wz[tid] = w;
    if (tid == 0) {
      stat[0] = mm;
      stat[1] = 1.0f / L;
    }
  }
  __syncthreads();
  const float inv_l = stat[1];
  if (a.mass && tid < a.T) {
    float sl = 0.f;
    for (int z = 0; z < a.ksplits; ++z) {
      if (wz[z] == 0.f) continue;
      const float* e = a.lslot + (((long)z * a.Npad + q) * a.T + tid) * 2;
      if (e[0] != 0.f) sl += e[0] * expf(e[1] - stat[0]);
    }
    a.mass[(long)q * a.T + tid] = sl * inv_l;
  }
  for (int c = tid * 4; c < a.ncols; c += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // 4 splits per step: their loads are unconditional and in flight together (an empty split's
    // partial is never written: its value is dropped by the select, not by a branch around the load)
    for (int z0 = 0; z0 < a.ksplits; z0 += 4) {
      float4 v[4];
      float w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int z = z0 + i < a.ksplits ? z0 + i : a.ksplits - 1;
        w[i] = z0 + i < a.ksplits ? wz[z] : 0.f;
        v[i] = *reinterpret_cast<const float4*>(a.part + ((long)z * a.Npad + q) * a.ncols + c);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool on = w[i] != 0.f;
        acc.x += w[i] * (on ? v[i].x : 0.f);
        acc.y += w[i] * (on ? v[i].y : 0.f);
        acc.z += w[i] * (on ? v[i].z : 0.f);
        acc.w += w[i] * (on ? v[i].w : 0.f);
      }
    }
    const float4 uu = *reinterpret_cast<const float4*>(a.U + (long)q * a.ldu + c);
    float4 g;
    g.x = acc.x * inv_l * uu.x;
    g.y = acc.y * inv_l * uu.y;
    g.z = acc.z * inv_l * uu.z;
    g.w = acc.w * inv_l * uu.w;
    *reinterpret_cast<float4*>(a.G + (long)q * a.ldg + c) = g;
  }
}

__device__ void read_combine_kernel(const rmem_read_combine_args& a, int) {
  __shared__ float sh[40];
  read_combine_body(a, blockIdx.x, sh);
}

struct Combine2Args {
  rmem_read_combine_args a, b;
};
__device__ void read_combine2_kernel(const Combine2Args& g, int) {
  __shared__ float sh[40];
  if ((int)blockIdx.x < g.a.N) read_combine_body(g.a, blockIdx.x, sh);
  else read_combine_body(g.b, blockIdx.x - g.a.N, sh);
}

static int read_combine_ok(const rmem_read_combine_args& a) {
  if (a.N <= 0 || a.Npad < a.N || a.T <= 0 || a.T > 64 || a.ksplits <= 0 || a.ksplits > 32) return 0;
  if ((a.ncols % 4) != 0 || !a.part || !a.ml || !a.U || !a.G || (a.ldu % 4) || (a.ldg % 4)) return 0;
  if (a.mass && !a.lslot) return 0;
  return 1;
}

extern "C" int rmem_attn_read_combine(const rmem_read_combine_args* ap, void* stream) {
  if (!ap || !read_combine_ok(*ap)) return RMEM_ERR_INVALID;
  return rmem::launch<rmem_read_combine_args, read_combine_kernel, 256>(*ap, dim3(ap->N), dim3(256), 0,
                                                                        static_cast<hipStream_t>(stream));
}

extern "C" int rmem_attn_read_combine2(const rmem_read_combine_args* ap, const rmem_read_combine_args* bp, void* stream) {
  if (!ap || !bp || !read_combine_ok(*ap) || !read_combine_ok(*bp)) return RMEM_ERR_INVALID;
  Combine2Args g{*ap, *bp};
  return rmem::launch<Combine2Args, read_combine2_kernel, 256>(g, dim3(ap->N + bp->N), dim3(256), 0,
                                                               static_cast<hipStream_t>(stream));
}
