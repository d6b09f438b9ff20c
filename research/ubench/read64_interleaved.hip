// NOT BUILT, NOT SHIPPED -- round-3 experiment kept for the record ("interleaved slices").
//
// Variant of csrc/read64.hip whose main pass folds the scores / weights of tile i + 1, cut into 16 slices, between the
// MFMAs of the 16 P.V steps of tile i (all 8 waves identical, one barrier per tile, V ring running through the tile
// boundaries).  Measured on MI355X (gpurun_out r03o / r03p, 480p K=4): loop per tile 8.9 k cycles against 9.75-10.25 k
// of the alternating-groups kernel that ships, but the exposed first-tile scores, nine scratch reloads in the window
// instantiation (14 k per tile) and the same launch tail left the whole reads no faster (bank ks=9 100.8 us vs 90-99,
// self 40.7 vs 35.5-38, window ks=2 82.9 vs 56-70), and one window geometry still aborted.  Lesson worth keeping: a
// counted wait tied to ring registers ("+v") must be ONE statement -- in two branches the tied input is copied before
// the wait in one of them, i.e. a register whose load has not landed is copied.
//
// read64.hip -- the memory read of GatedPropagation / LocalGatedPropagation (layers/attention.py:
// 174-209, 289-358; call sites layers/transformer.py:1183, 1199, 1227) as ONE flash-style launch per
// read: S = Q.K^T, softmax and O = P.V without ever writing the probability matrix to HBM.
//
// Work unit = (64-query tile, ALL 1024 columns of [V | ID_V], key split).  One workgroup = 8 waves,
// TWO per SIMD with 256 registers each: wave w owns the 64 queries x 128 columns [128 w, 128 w + 128)
// of O = 2 x 4 MFMA tiles = 128 accumulator registers.  (Round 2's unit was 128 queries x a 512-column
// half on 4 waves with the whole register file each: Q.K^T was computed once per column half, and
// with one wave per SIMD the matrix pipe idled through every score / softmax phase -- 24-29 % busy,
// profiles/r02_x_pmc_read.json.)
//
// Per 64-key tile a wave does two things that touch disjoint buffers and can therefore run in either
// order inside ONE barrier interval:
//   SCORE(i+1): S^T = K.Q^T for 16 queries x 32 keys (wave w: query group w & 3, key half w >> 2) on
//       v_mfma_f32_16x16x32_f16, K and Q fragments from LDS; "swapped" (rows = keys) so that a query's
//       scores live in 4 lanes; weights P = 2^(y - m) against a FIXED per-query reference m go to the
//       other P image as fp16 hi / lo planes;
//   PV(i): O += P.V for all 64 queries x this wave's 128 columns on v_mfma_f32_32x32x16_f16: A = P
//       fragments from LDS (shared by the eight waves), B = V fragments straight from global memory
//       ("blocked-16" layout: one contiguous KiB per load instruction).
// Waves 0-3 run SCORE then PV, waves 4-7 PV then SCORE: wave w and wave w + 4 share a SIMD, so each
// SIMD always has one wave in the matrix-heavy P.V phase while the other does LDS reads, exp2 and
// conversions.  K tiles arrive by LDS-DMA (global_load_lds_dwordx4, two tiles ahead, XOR swizzle applied
// to the SOURCE address: no staging registers, no ds_write phase); P and K are double-buffered.
//
// The fixed reference m is the row maximum over the unit's keys from a first pass with the hi planes only
// (8 small MFMAs per wave per tile; error a few hundredths), so the weights stay within a few per cent of 1
// at the maximum and the accumulators never need a rescale.  Exactness for ANY input: the main
// pass flags a score above m + RD_THR (weights would leave fp16); a flagged unit is simply redone with the
// reference taken from the exact three-product scores of every key (bit-identical to the main pass's, so
// the weights are <= 1).  Not taken on real data; tests/test_hip_ops.py forces it.
//
// Split precision: every product is hi*lo' + lo*hi' + hi*hi' (fp32 accumulate).  Key splits write
// un-normalised partials + (max, sum) [+ per-slot (sum, max)]; rmem_attn_read_combine merges them.
#include "../../include/rmem_hip.h"
#include "gemm_core.h"
#include "attn_common.h"
#include "launch.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

#define RMEM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// LDS map (the images' internal layouts are described where their addresses are formed, read64_body).
constexpr int R6_Q = 0;                     // [plane][d-step][64 queries][64 B]
constexpr int R6_K = 32768;                 // 2 buffers x [plane][d-step][64 keys][64 B]
constexpr int R6_P = 98304;                 // 2 buffers x [plane][k-step][64 queries][32 B]
constexpr int R6_SL = 131072;               // per-slot sums [16 slots][2 key halves][64 queries] fp32
constexpr int R6_MX = R6_SL + 8192;         // [2 key halves][64 queries] row maximum (reference pass)
constexpr int R6_L = R6_MX + 512;           // [2][64] row sums
constexpr int R6_FL = R6_L + 512;           // [8] overflow flags
constexpr int R6_DUMMY = R6_FL + 64;        // 1 KiB nobody reads: target of the K requests that fetch no tile
constexpr int R6_LDS = R6_DUMMY + 1024;
constexpr float RD_NEG = -3.0e38f;
constexpr float RD_THR = 14.0f;             // log2 domain: weights up to 2^14 = 16384 < 65504 (fp16 hi plane)

#define R6_OPAQUE(x) asm volatile("" : "+v"(x))

// VAR (experiments, tracing kernel only): 8 = never wait for V fragments (timing only: results are wrong).
template <int TRACE, int VAR, int MODE>
__device__ __forceinline__ void read64_mode(const rmem_read_args& a, const int blk, char* smem, long long* trace_base) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qg = wave & 3, kh = wave >> 2;            // score role: query group (16 queries), key half (32 keys)
  const int jq = lane & 15, lb = lane >> 4;           // 16x16 tile: column (query) / row block (4 keys)
  const int j = lane & 31, hi = lane >> 5;            // 32x32 tile (P.V)

  // ---- work unit.  Units are ordered (split, query tile); every XCD (block b runs on XCD b % 8:
  // observed placement, used for speed only) owns a contiguous chunk, so the units of one XCD share a
  // key split, i.e. the same K / V bytes in that XCD's L2.
  const int nq = (a.N + 63) / 64;
  const int nunits = nq * a.ksplits;
  const int chunk = (nunits + 7) / 8;
  const int jj = blk >> 3;
  const int u = (blk & 7) * chunk + jj;
  if (jj >= chunk || u >= nunits) return;
  const int qtile = u % nq;
  const int z = u / nq;

  const int tv = (a.N + 63) / 64;            // 64-key tiles of a slot that hold valid keys
  int k_lo, k_hi;
  if (MODE == 0) {
    k_lo = 0;
    k_hi = a.T * tv;
  } else {                                   // key tiles visible under the 15x15 window to this query tile
    const int q_lo = qtile * 64;
    const int q_hi = q_lo + 63 < a.N - 1 ? q_lo + 63 : a.N - 1;
    int y_lo = q_lo / a.w - 7, y_hi = q_hi / a.w + 7;
    if (y_lo < 0) y_lo = 0;
    if (y_hi > a.h - 1) y_hi = a.h - 1;
    k_lo = (y_lo * a.w) / 64;
    k_hi = ((y_hi + 1) * a.w + 63) / 64;
  }
  const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
  const int lo = k_lo + z * per;
  const int hi_t = lo + per < k_hi ? lo + per : k_hi;
  const int n = hi_t - lo;

  const int q = qtile * 64 + qg * 16 + jq;            // this lane's query in the score phase
  const bool qvalid = q < a.N;
  if (n <= 0) {                                       // no key tile in this split (narrow band)
    if (tid < 64) {
      float* mlp = a.ml + ((long)z * a.Npad + qtile * 64 + tid) * 2;
      mlp[0] = RD_NEG;
      mlp[1] = 0.f;
    }
    return;
  }
  long long* trace = (TRACE && trace_base) ? trace_base + (long)blk * 64 : nullptr;   // debug aid: shader-clock stamps per block
  float* lslot_out = a.lslot;
  long long tacc[4] = {0, 0, 0, 0};                     // TRACE: this wave's cycles in SCORE / PV / waiting at barriers
  if (trace && tid == 0) trace[0] = __builtin_readcyclecounter();

  SlotLut lut;
  lut.load(a.slot_map, MODE == 0 ? a.T : 1);

  float* sl_sum = reinterpret_cast<float*>(smem + R6_SL);
  float* mx_ex = reinterpret_cast<float*>(smem + R6_MX);
  float* l_ex = reinterpret_cast<float*>(smem + R6_L);
  int* flag = reinterpret_cast<int*>(smem + R6_FL);

  // ---- LDS addresses: ONE opaque register per image; everything that varies inside the loops (d-step, k-step,
  // key group, plane, buffer) is an ADDITIVE constant that folds into the 16-bit offset field of the DS
  // instruction.  (Round-3 history: with the XOR swizzle spanning the d-step bits, each d-step needed its own
  // address -- eight registers per family, hoisted out of the loop, spilled and reloaded inside the P.V cluster,
  // where every compiler-visible vector-memory access drains the V ring, s_waitcnt vmcnt(0): 6.5 k instead of 4.2 k
  // cycles per tile; re-deriving them at the use cost ~80 VALU instructions per tile on an issue-bound loop.)
  //
  // Q / K images: [d-step k4 (4)][row (64)][64 B]; the 16-byte chunk lb of a (k4, row) segment sits at slot
  // lb ^ f(row >> 2), f = (0, 3, 2, 1).  A 16x16x32 fragment read (row = lane & 15, chunk = lane >> 4) is served in
  // groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, ..: within a group the (row & 3, slot) pairs are all
  // different, i.e. 16 distinct 16-byte slots of the 256-byte bank row: conflict-free.
  // P image: [k-step ks (4)][row (64)][32 B]; chunk hi of a (ks, row) segment at slot hi ^ ((row >> 3) & 1):
  // conflict-free for the 32x32x16 A-fragment reads (row = lane & 31, chunk = lane >> 5).
  const int fsw = (4 - ((jq >> 2) & 3)) & 3;                          // f of this lane's score rows (rows = .. + jq)
  int ak0 = R6_K + (kh * 32 + jq) * 64 + ((lb ^ fsw) << 4);           // K row kh*32 (+ kt*16) + jq; + k4*4096 + kt*1024
  int aq0 = R6_Q + (qg * 16 + jq) * 64 + ((lb ^ fsw) << 4);           // Q row qg*16 + jq;           + k4*4096
  R6_OPAQUE(ak0);
  R6_OPAQUE(aq0);
  auto ak = [&](int k4) __attribute__((always_inline)) { return ak0 + k4 * 4096; };
  auto aq = [&](int k4) __attribute__((always_inline)) { return aq0 + k4 * 4096; };
  int apw0;                                           // P stores: row qg*16 + jq, keys kh*32 + kt*16 + lb*4 .. +3 = k-step 2 kh + kt
  {
    const int row = qg * 16 + jq;
    apw0 = R6_P + kh * 4096 + row * 32 + (((lb >> 1) ^ ((row >> 3) & 1)) << 4) + (lb & 1) * 8;
    R6_OPAQUE(apw0);
  }
  auto apw = [&](int kt) __attribute__((always_inline)) { return apw0 + kt * 2048; };
  int apr0 = R6_P + j * 32 + ((hi ^ ((j >> 3) & 1)) << 4);            // P fragments: row j (+ 32 qi), chunk hi; + ks*2048 + qi*1024
  R6_OPAQUE(apr0);
  auto apr = [&](int ks) __attribute__((always_inline)) { return apr0 + ks * 2048; };
  // Staging (LDS-DMA and the register-staged reference pass): a wave-instruction moves 1 KiB = 16 rows x 64 B of
  // one d-step block; lane L lands at row L >> 2, slot L & 3 of the piece and therefore fetches source chunk
  // (L & 3) ^ f(L >> 4) of its row.  A [64 rows][256 B] plane tile is 16 pieces (d-step k4 = P >> 2, row group P & 3);
  // wave w moves pieces w and w + 8: the same rows, d-steps w >> 2 and (w >> 2) + 2 (source + 128 B, image + 8 KB).
  int dma_off0;
  {
    const int rowp = lane >> 2, sl = lane & 3;
    const int g = wave & 3, k4 = wave >> 2;
    dma_off0 = (16 * g + rowp) * 256 + k4 * 64 + ((sl ^ ((4 - ((rowp >> 2) & 3)) & 3)) << 4);
    R6_OPAQUE(dma_off0);
  }
  const int dma_dst0 = (wave >> 2) * 4096 + (wave & 3) * 1024;        // image offset of piece `wave`
  // (inline assembly: the builtin form makes the compiler wait for the transfer -- s_waitcnt vmcnt(0) -- before the
  // next vector-memory instruction; this way the only waits are the explicit ones in front of the barriers.  The
  // compiler does not count these requests: its own vmcnt waits can only become stricter, never too weak.)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto dma_plane = [&](const h16_t* plane_rows, int lds_base) __attribute__((always_inline)) {
    const char* src = reinterpret_cast<const char*>(plane_rows);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + lds_base + dma_dst0 + pc * 8192);
      const char* g = src + dma_off0 + pc * 128;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g) : "memory");
    }
  };
  // Key tile index -> (logical slot, key offset inside the slot).  One division per pass; the loops then step
  // (a division per use was ~60 scalar instructions, three times per tile)
  struct TileIter {
    int t, kt;
    long kslot, vslot;                                // element offsets of the slot's K / V planes
    __device__ __forceinline__ int key0() const { return kt * 64; }
  };
  auto tinit = [&](TileIter& ti, int i) __attribute__((always_inline)) {
    ti.t = 0;
    ti.kt = i;
    if (MODE == 0) {
      ti.t = i / tv;
      ti.kt = i - ti.t * tv;
    }
    const long sl = lut(ti.t);
    ti.kslot = sl * a.k_slot_stride;
    ti.vslot = sl * a.v_slot_stride;
  };
  auto tstep = [&](TileIter& ti) __attribute__((always_inline)) {
    ++ti.kt;
    if (MODE == 0 && ti.kt == tv) {                   // next slot (once per 27 tiles at 480p): the only place a slot is looked up
      ti.kt = 0;
      ++ti.t;
      const long sl = lut(ti.t);
      ti.kslot = sl * a.k_slot_stride;
      ti.vslot = sl * a.v_slot_stride;
    }
  };
  auto dma_k = [&](const TileIter& ti, int buf, bool both) __attribute__((always_inline)) {
    const long base = ti.kslot + (long)ti.key0() * 128;
    dma_plane(a.kh + base, R6_K + buf * 32768);
    if (both) dma_plane(a.kl + base, R6_K + buf * 32768 + 16384);
  };

  // ---- Q tile -> LDS (once)
  dma_plane(a.qh + (long)qtile * 64 * 128, R6_Q);
  dma_plane(a.ql + (long)qtile * 64 * 128, R6_Q + 16384);

  const float sl2e = a.scale * 1.44269504088896341f;
  constexpr float LN2 = 0.693147180559945f;
  int qy = 0, qx = 0;
  const float inv_w = MODE == 1 ? 1.0f / (float)a.w : 0.f;   // (a.w >= 1 is validated for mode 1)
  const float* Rq = nullptr;
  const int rcs = a.rcs > 0 ? a.rcs : 1;
  if (MODE == 1) {
    qy = fast_div(qvalid ? q : 0, inv_w);
    qx = (qvalid ? q : 0) - qy * a.w;
    Rq = a.R + (long)(qvalid ? q : 0) * a.ldr;
  }
  // Windowed read: relative bias (log2 domain) of this lane's 8 keys of the tile at key0, RD_NEG where the key
  // is outside the 15x15 window / the image.  All gathers are issued unconditionally (index 0 where masked)
  // and back to back; one division per 4 consecutive keys.
  auto window_terms = [&](int key0, float (&rb)[8]) __attribute__((always_inline)) {
    int idx[8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int tok0 = key0 + kh * 32 + kt * 16 + lb * 4;
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
        idx[kt * 4 + e] = valid ? ((dy + 7) * 15 + dx + 7) * rcs : -1;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) rb[r] = Rq[idx[r] < 0 ? 0 : idx[r]];
#pragma unroll
    for (int r = 0; r < 8; ++r) rb[r] = idx[r] < 0 ? RD_NEG : rb[r] * 1.44269504088896341f;
  };

  // ---- scores of one tile for this lane: y[kt*4 + r] = log2-domain logit of key kh*32 + kt*16 + lb*4 + r
  // (RD_NEG where masked).  EXACT: the three-product form (main pass, redo reference pass); else the hi planes only.
  // kb = byte offset of the tile's K image relative to R6_K.
  int cur_t = -1;
  float bias2 = 0.f;
  auto scores = [&](const TileIter& ti, int kb, auto EX, float (&y)[8]) __attribute__((always_inline)) {
    constexpr bool EXACT = decltype(EX)::value;
    const int t = ti.t, key0 = ti.key0();
    if (MODE == 0 && t != cur_t) {
      cur_t = t;
      bias2 = ((a.bias && qvalid) ? a.bias[(long)q * a.T + t] : 0.f) * sl2e;     // (padding queries: no row in bias)
    }
    f32x4_t s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] = 0.f;
    // fragments of d-step k4 + 1 are requested before the MFMAs of step k4 (two register sets): read one step at
    // a time the phase was four LDS round trips long (~3.4 k cycles per tile beside the partner wave's P.V)
    constexpr int NF = EXACT ? 6 : 3;                 // [0] Q hi, [1] K hi (keys 0-15), [2] K hi (16-31), [3..5] the lo planes
    frag8_t fa[NF], fb[NF];
    auto fload = [&](frag8_t (&f)[NF], auto K4) __attribute__((always_inline)) {
      constexpr int k4 = decltype(K4)::value;
      f[0] = *reinterpret_cast<const frag8_t*>(smem + aq(k4));
      f[1] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb);
      f[2] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + 1024);
      if constexpr (EXACT) {
        f[3] = *reinterpret_cast<const frag8_t*>(smem + aq(k4) + 16384);
        f[4] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + 16384);
        f[5] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + (16384 + 1024));
      }
    };
    fload(fa, std::integral_constant<int, 0>{});
    static_for<4>([&](auto K4) {
      constexpr int k4 = K4.value;
      frag8_t (&c)[NF] = (k4 & 1) ? fb : fa;
      if constexpr (k4 < 3) fload((k4 & 1) ? fa : fb, std::integral_constant<int, (k4 < 3 ? k4 + 1 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (EXACT) {
        s[0] = RMEM_MFMA16(c[1], c[3], s[0]);         // small terms first: K hi . Q lo, K lo . Q hi, K hi . Q hi
        s[1] = RMEM_MFMA16(c[2], c[3], s[1]);
        s[0] = RMEM_MFMA16(c[4], c[0], s[0]);
        s[1] = RMEM_MFMA16(c[5], c[0], s[1]);
      }
      s[0] = RMEM_MFMA16(c[1], c[0], s[0]);
      s[1] = RMEM_MFMA16(c[2], c[0], s[1]);
      __builtin_amdgcn_sched_barrier(0);
    });
    const bool padded = key0 + 64 > a.N;
    if (MODE == 0 && !padded) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[kt * 4 + r] = fmaf(s[kt][r], sl2e, bias2);
    } else if (MODE == 0) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = key0 + kh * 32 + kt * 16 + lb * 4 + r;
          y[kt * 4 + r] = tok < a.N ? fmaf(s[kt][r], sl2e, bias2) : RD_NEG;
        }
    } else {
      float rb[8];
      window_terms(key0, rb);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float rbv = rb[kt * 4 + r];
          y[kt * 4 + r] = rbv > -2.9e38f ? fmaf(s[kt][r], sl2e, rbv) : RD_NEG;
        }
    }
    return t;
  };

  int vcol = (wave * 128 + j) * 16 + hi * 8;           // V fragments: lane = column, 8 consecutive keys (16 B)
  R6_OPAQUE(vcol);

  float m = RD_NEG;                                   // this query's reference (log2 domain)
  for (int attempt = 0; attempt < 2; ++attempt) {
    // ================= reference pass: m <= (and close to) the maximum of this query's scores over the unit's keys
    {
      float mest = RD_NEG;
      cur_t = -1;
      __syncthreads();                                // (redo: every wave is done with the images of the main pass)
      if (attempt == 0) {
        // First attempt: the hi planes only (8 small MFMAs per wave per tile; the maximum is off by a few hundredths).
        // Tiles go in PAIRS, one barrier per pair.  The K tiles of this pass are staged through REGISTERS (4 loads +
        // 4 ds_write_b128 per wave per pair, swizzle on the source address, the same image the LDS-DMA builds): the
        // DMA path moves ~12 B / clk / CU, and at 16 KB per tile that alone was 1.3-1.9 k cycles per tile of a pass
        // whose arithmetic is nothing.  (Every key is looked at: a sub-sampled maximum misses the one dominant key
        // of peaked attention -- the diagonal of the self read, the same position of the previous frame -- and every
        // such unit then pays the exact redo: measured, LSTT forward 0.94 -> 1.07 ms.)
        constexpr int RP = 3;                          // ring of 3 pairs = 6 hi-plane tiles in [R6_K, R6_K + 96 KB)
        const int npairs = (n + 1) / 2;
        TileIter tld, tcmp;
        tinit(tld, lo);
        tcmp = tld;
        u32x4_t rk[4];                                 // [tile of the pair][piece wave / wave + 8]
        auto kload = [&]() __attribute__((always_inline)) {             // request the pair at tld (clamped to valid tiles)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const char* src = reinterpret_cast<const char*>(a.kh + tld.kslot + (long)tld.key0() * 128) + dma_off0;
            rk[e * 2 + 0] = *reinterpret_cast<const u32x4_t*>(src);
            rk[e * 2 + 1] = *reinterpret_cast<const u32x4_t*>(src + 128);
            if (tld.t * tv + tld.kt + 1 < k_hi) tstep(tld);             // (never steps past the last tile of the read:
                                                                        //  a clamped request repeats a tile nobody uses)
          }
        };
        auto kstore = [&](int pair) __attribute__((always_inline)) {
          const int base = R6_K + (pair % RP) * 32768 + dma_dst0 + lane * 16;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            *reinterpret_cast<u32x4_t*>(smem + base + e * 16384) = rk[e * 2 + 0];
            *reinterpret_cast<u32x4_t*>(smem + base + e * 16384 + 8192) = rk[e * 2 + 1];
          }
        };
        kload();
        kstore(0);
        if (npairs > 1) kload();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the Q tile's DMA)
        __syncthreads();
#pragma clang loop unroll(disable)
        for (int p = 0; p < npairs; ++p) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int x = 2 * p + e;
            if (x < n) {
              float y[8];
              scores(tcmp, (p % RP) * 32768 + e * 16384, std::false_type{}, y);
              tstep(tcmp);
#pragma unroll
              for (int r = 0; r < 8; ++r) mest = fmaxf(mest, y[r]);
            }
          }
          if (p + 1 < npairs) {
            kstore(p + 1);                             // requested one iteration ago; its slot was last read in iteration p - 2
            if (p + 2 < npairs) kload();
          }
          __syncthreads();
        }
      } else {
        // Redo: the exact three-product scores of EVERY key (bit-identical to the main pass's), both planes through
        // the two K buffers, one tile per barrier.
        TileIter tdma, tcmp;
        tinit(tdma, lo);
        tcmp = tdma;
        int issued = 0;                               // tiles requested so far
        auto request = [&](int upto) __attribute__((always_inline)) {
          for (; issued < upto && issued < n; ++issued) {
            dma_k(tdma, issued & 1, true);
            tstep(tdma);
          }
        };
        request(2);
#pragma clang loop unroll(disable)
        for (int x = 0; x < n; ++x) {
          if (issued - (x + 1) >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // 4 pieces per tile per wave
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();                            // tile x landed for every wave; the buffer of tile x - 1 is free
          if (x >= 1) request(x + 2);
          float y[8];
          scores(tcmp, (x & 1) * 32768, std::true_type{}, y);
          tstep(tcmp);
#pragma unroll
          for (int r = 0; r < 8; ++r) mest = fmaxf(mest, y[r]);
        }
      }
      mest = fmaxf(mest, __shfl_xor(mest, 16));
      mest = fmaxf(mest, __shfl_xor(mest, 32));
      __syncthreads();                                // all fragment reads of the pass are done: the K / P regions are free
      if (lb == 0) mx_ex[kh * 64 + qg * 16 + jq] = mest;
      // per-slot sums and flags start from zero
      for (int e = tid; e < 2048; e += 512) sl_sum[e] = 0.f;
      if (tid < 8) flag[tid] = 0;
      // first K tiles of the main pass
      TileIter t01;
      tinit(t01, lo);
      dma_k(t01, 0, true);
      tstep(t01);
      if (n > 1) dma_k(t01, 1, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      m = fmaxf(mx_ex[qg * 16 + jq], mx_ex[64 + qg * 16 + jq]);
    }
    if (trace && tid == 0) trace[1] = __builtin_readcyclecounter();

    // ================= main pass
    f32x16_t o[2][4];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qi][ci][r] = 0.f;
    float l = 0.f, lcur = 0.f;                        // row sum of the finished slots / of the current slot
    int sum_t = -1;                                   // slot lcur belongs to
    float ytop = RD_NEG;                              // largest logit seen (overflow test at the end of the pass)
    cur_t = -1;

    // ---- SCORE(i) in 16 slices.  Every wave runs the P.V of tile i and, BETWEEN its MFMAs, the scores / weights of
    // tile i + 1 (other K and P buffers): one slice after the 6 MFMAs of each of the 16 P.V steps, a handful of
    // instructions per MFMA gap.  (The first form of this kernel alternated whole phases between the two waves of a
    // SIMD -- one in P.V while the other scored.  Measured per tile: P.V 5.3 k cycles beside a scoring partner against
    // 3.1 k of MFMA time, 10 k per tile; two waves that both feed the matrix pipe keep it 90 % busy, so the small work
    // is folded into the stream that feeds it.)  Slices 0-8: fragments and MFMAs of the 4 d-steps x 2 key groups (16
    // registers of fragments live); 9: logits, masks, overflow test; 10-13: weights; 14-15: fp16 hi / lo planes -> P.
    f32x4_t sc[2];
    frag8_t fq[2], fk[2];                             // Q hi / lo of a d-step; K hi / lo of a (d-step, key group)
    float y[8];                                       // logits, overwritten by the weights
    auto score_slice = [&](auto S, const TileIter& ti, int kb, int pbuf) __attribute__((always_inline)) {
      constexpr int sl = decltype(S)::value;
      auto qload = [&](auto K4) __attribute__((always_inline)) {
        constexpr int k4 = decltype(K4)::value;
        fq[0] = *reinterpret_cast<const frag8_t*>(smem + aq(k4));
        fq[1] = *reinterpret_cast<const frag8_t*>(smem + aq(k4) + 16384);
      };
      auto kload = [&](auto K4, auto KT) __attribute__((always_inline)) {
        constexpr int k4 = decltype(K4)::value, kt = decltype(KT)::value;
        fk[0] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + kt * 1024);
        fk[1] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + (16384 + kt * 1024));
      };
      auto fmul = [&](auto KT) __attribute__((always_inline)) {       // small terms first: K hi . Q lo, K lo . Q hi, K hi . Q hi
        constexpr int kt = decltype(KT)::value;
        sc[kt] = RMEM_MFMA16(fk[0], fq[1], sc[kt]);
        sc[kt] = RMEM_MFMA16(fk[1], fq[0], sc[kt]);
        sc[kt] = RMEM_MFMA16(fk[0], fq[0], sc[kt]);
      };
      if constexpr (sl == 0) {
        const int t = ti.t;
        if (MODE == 0 && t != cur_t) {
          cur_t = t;
          bias2 = ((a.bias && qvalid) ? a.bias[(long)q * a.T + t] : 0.f) * sl2e;     // (padding queries: no row in bias)
        }
        if (t != sum_t) {
          if (sum_t >= 0) {                           // (wave-uniform) slot finished: park its sum
            float v = lcur + __shfl_xor(lcur, 16);
            v += __shfl_xor(v, 32);
            if (lb == 0) sl_sum[(sum_t * 2 + kh) * 64 + qg * 16 + jq] = v;
          }
          l += lcur;
          lcur = 0.f;
          sum_t = t;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[kt][r] = 0.f;
        qload(std::integral_constant<int, 0>{});
        kload(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      } else if constexpr (sl >= 1 && sl <= 8) {      // multiply pair (sl - 1), fetch pair sl
        constexpr int done = sl - 1, next = sl;
        fmul(std::integral_constant<int, (done & 1)>{});
        if constexpr (next < 8) {
          if constexpr ((next & 1) == 0) qload(std::integral_constant<int, (next < 8 ? next >> 1 : 0)>{});
          kload(std::integral_constant<int, (next < 8 ? next >> 1 : 0)>{}, std::integral_constant<int, (next & 1)>{});
        }
      } else if constexpr (sl == 9) {
        const int key0 = ti.key0();
        const bool masked = MODE == 1 || key0 + 64 > a.N;     // (wave-uniform) the tile may hold RD_NEG sentinels
        if (MODE == 0 && !masked) {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[kt * 4 + r] = fmaf(sc[kt][r], sl2e, bias2);
        } else if (MODE == 0) {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int tok = key0 + kh * 32 + kt * 16 + lb * 4 + r;
              y[kt * 4 + r] = tok < a.N ? fmaf(sc[kt][r], sl2e, bias2) : RD_NEG;
            }
        } else {
          float rb[8];
          window_terms(key0, rb);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float rbv = rb[kt * 4 + r];
              y[kt * 4 + r] = rbv > -2.9e38f ? fmaf(sc[kt][r], sl2e, rbv) : RD_NEG;
            }
        }
        // the largest score so far (sentinels are far below): tested against the reference once, after the pass
        ytop = fmaxf(ytop, fmaxf(fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])), fmaxf(fmaxf(y[4], y[5]), fmaxf(y[6], y[7]))));
      } else if constexpr (sl >= 10 && sl <= 13) {
        constexpr int e0 = 2 * (sl - 10);
        const bool masked = MODE == 1 || ti.key0() + 64 > a.N;
        if (!masked) {
#pragma unroll
          for (int e = e0; e < e0 + 2; ++e) {
            y[e] = __builtin_amdgcn_exp2f(y[e] - m);
            lcur += y[e];
          }
        } else {
          asm volatile("" ::: "memory");              // keeps this form a branch (as a select it costs every tile two more VALU per key)
#pragma unroll
          for (int e = e0; e < e0 + 2; ++e) {
            float p = __builtin_amdgcn_exp2f(y[e] - m);       // sentinels (-3e38) give exactly 0 unless m is one too
            p = y[e] > -2.9e38f ? p : 0.f;
            y[e] = p;
            lcur += p;
          }
        }
      } else if constexpr (sl == 14 || sl == 15) {
        constexpr int kt = sl - 14;
        // hi = fp16(p), lo = fp16(p - hi): packed conversions (v_cvt_pk_f16_f32, round to nearest even)
        f32x2_t p0, p1;
        p0[0] = y[kt * 4 + 0];
        p0[1] = y[kt * 4 + 1];
        p1[0] = y[kt * 4 + 2];
        p1[1] = y[kt * 4 + 3];
        const f16x2_t h0 = __builtin_convertvector(p0, f16x2_t), h1 = __builtin_convertvector(p1, f16x2_t);
        const f32x2_t r0 = p0 - __builtin_convertvector(h0, f32x2_t), r1 = p1 - __builtin_convertvector(h1, f32x2_t);
        const f16x2_t l0 = __builtin_convertvector(r0, f16x2_t), l1 = __builtin_convertvector(r1, f16x2_t);
        u32x2_t wh, wl;
        wh[0] = __builtin_bit_cast(uint32_t, h0);
        wh[1] = __builtin_bit_cast(uint32_t, h1);
        wl[0] = __builtin_bit_cast(uint32_t, l0);
        wl[1] = __builtin_bit_cast(uint32_t, l1);
        *reinterpret_cast<u32x2_t*>(smem + apw(kt) + pbuf * 16384) = wh;
        *reinterpret_cast<u32x2_t*>(smem + apw(kt) + pbuf * 16384 + 8192) = wl;
      }
    };

    // ---- V fragments: a ring of 4 steps (step = k-step x 32-column tile, 8 registers) that runs THROUGH the tile
    // boundaries.  Requested by inline assembly four steps ahead and awaited with COUNTED vmcnt tied to the ring
    // registers ("+v"): left to the compiler, its wait-count pass protects the ring with s_waitcnt vmcnt(0), i.e.
    // waits for the K transfer that was just requested.  The compiler sees no vector-memory load in the loop.
    u32x4_t vr[4][2];
    const h16_t* vhp = a.vh;                          // blocked-16 planes of the tile whose P.V runs / of the next tile
    const h16_t* vlp = a.vl;
    const h16_t* vhn = a.vh;
    const h16_t* vln = a.vl;
    auto vreq = [&](auto S) __attribute__((always_inline)) {     // step S of this tile (S < 16) or S - 16 of the next
      constexpr int sidx = decltype(S)::value;
      constexpr int ks = (sidx & 15) >> 2, ci = sidx & 3;
      // scalar base + 32-bit lane offset (k-step: 16 keys x 1024 columns = 32 KB) + immediate (32-column tile: 1 KB)
      const h16_t* bh = sidx < 16 ? vhp : vhn;
      const h16_t* bl = sidx < 16 ? vlp : vln;
      const int off = vcol * 2 + ks * 32768;
      u32x4_t& d0 = vr[sidx & 3][0];                  // (asm operands do not capture: name the slots first)
      u32x4_t& d1 = vr[sidx & 3][1];
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d0) : "v"(off), "s"(bh), "n"(ci * 1024));
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d1) : "v"(off), "s"(bl), "n"(ci * 1024));
    };
    // Vector-memory requests in flight, oldest first, when step s waits for its fragments: V(s) .. V(s+3) of the ring
    // and, for s < 4, the 4 K-transfer pieces issued at the top of the iteration.  Both are ALWAYS issued -- a 1 KiB
    // dummy target for the K pieces when there is no tile to fetch, a re-read of the current planes for the V steps
    // past the last tile -- so that each wait is ONE statement with a constant count: vmcnt(10) for s < 4, vmcnt(6)
    // after.  (A wait that sits in two branches -- "if (more) vmcnt(6) else vmcnt(2)" -- makes the ring registers a
    // phi; the compiler then copies the tied operand BEFORE the wait in one of the branches, i.e. copies a register
    // whose load has not landed: wrong, run-to-run varying results on the last tile of a unit.)
    auto vwait = [&](auto S) __attribute__((always_inline)) {
      constexpr int sidx = decltype(S)::value;
      u32x4_t& d0 = vr[sidx & 3][0];
      u32x4_t& d1 = vr[sidx & 3][1];
      if constexpr (VAR & 8) {                        // (timing experiment of the tracing kernel: never wait; wrong results)
        asm volatile("" : "+v"(d0), "+v"(d1));
      } else if constexpr (sidx < 4) {
        asm volatile("s_waitcnt vmcnt(10)" : "+v"(d0), "+v"(d1));
      } else {
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(d0), "+v"(d1));
      }
    };

    TileIter t_sc, t_dma, t_v;                        // tiles of the next SCORE, K request, V base
    tinit(t_sc, lo);
    tinit(t_dma, lo + 2 < hi_t ? lo + 2 : lo);
    tinit(t_v, lo);
    // weights of the first tile; V fragments of its first four steps
    static_for<16>([&](auto S) { score_slice(S, t_sc, 0, 0); });
    tstep(t_sc);
    {
      const long vb = t_v.vslot + (long)(t_v.key0() >> 4) * (1024 * 16);
      vhp = a.vh + vb;
      vlp = a.vl + vb;
      if (n > 1) tstep(t_v);
    }
    vreq(std::integral_constant<int, 0>{});
    vreq(std::integral_constant<int, 1>{});
    vreq(std::integral_constant<int, 2>{});
    vreq(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int it = 0; it < n; ++it) {
      const bool more = it + 1 < n;
      long long t0 = 0;
      if (TRACE) t0 = __builtin_readcyclecounter();
      // K(it + 2) into the buffer of K(it), whose readers (SCORE(it), previous iteration) are past the barrier; when
      // there is no such tile the 4 pieces go to a 1 KiB dummy target (keeps the request count of vwait() constant)
      {
        const bool real = it + 2 < n;
        const long base = t_dma.kslot + (long)t_dma.key0() * 128;
        const int dst = real ? R6_K + (it & 1) * 32768 + dma_dst0 : R6_DUMMY;
        const int dstep = real ? 8192 : 0;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + dst + pl * (real ? 16384 : 0) + pc * dstep);
            const char* g = reinterpret_cast<const char*>((pl ? a.kl : a.kh) + base) + dma_off0 + pc * 128;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(g) : "memory");
          }
        if (real) tstep(t_dma);
      }
      if (more) {                                     // V planes of the next tile (its first steps are requested below)
        const long vb = t_v.vslot + (long)(t_v.key0() >> 4) * (1024 * 16);
        vhn = a.vh + vb;
        vln = a.vl + vb;
        if (it + 2 < n) tstep(t_v);
      }
      const int pbuf = it & 1, kb = ((it + 1) & 1) * 32768, pnext = (it + 1) & 1;
      frag8_t pf[4];                                  // P fragments [qi][plane] of one k-step (fetched right after the
      auto pload = [&](auto KS) __attribute__((always_inline)) {      //  last MFMA of the previous k-step issues)
        constexpr int ks = decltype(KS)::value;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            pf[qi * 2 + p] = *reinterpret_cast<const frag8_t*>(smem + apr(ks) + (pbuf * 16384 + p * 8192 + qi * 1024));
      };
      pload(std::integral_constant<int, 0>{});
      static_for<16>([&](auto S) {
        constexpr int sidx = S.value;
        constexpr int ks = sidx >> 2, ci = sidx & 3;
        __builtin_amdgcn_sched_barrier(0);
        vwait(S);
        const frag8_t vh = __builtin_bit_cast(frag8_t, vr[sidx & 3][0]);
        const frag8_t vl = __builtin_bit_cast(frag8_t, vr[sidx & 3][1]);
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {              // small terms first
          o[qi][ci] = RMEM_MFMA(pf[qi * 2 + 0], vl, o[qi][ci]);
          o[qi][ci] = RMEM_MFMA(pf[qi * 2 + 1], vh, o[qi][ci]);
          o[qi][ci] = RMEM_MFMA(pf[qi * 2 + 0], vh, o[qi][ci]);
        }
        if constexpr (ci == 3 && ks < 3) pload(std::integral_constant<int, (ks < 3 ? ks + 1 : 0)>{});
        vreq(std::integral_constant<int, sidx + 4>{});       // (steps 16-19: the next tile's first four, see vwait)
        if (more) score_slice(S, t_sc, kb, pnext);
        __builtin_amdgcn_sched_barrier(0);
      });
      if (more) {
        tstep(t_sc);
        vhp = vhn;
        vlp = vln;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // all but the next tile's first four V steps: the K transfer is in
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      if (TRACE) { const long long t1 = __builtin_readcyclecounter(); tacc[1] += t1 - t0; t0 = t1; }
      __syncthreads();                                // P(it + 1), K(it + 2) visible; P(it), K(it + 1) may be overwritten
      if (TRACE) tacc[2] += __builtin_readcyclecounter() - t0;
    }
    if (trace && tid == 0) trace[2] = __builtin_readcyclecounter();

    // ---- overflow?  (wave-uniform flags; the unit is redone against the exact reference)
    if (attempt == 0) {
      const int any_over = __any(ytop - m > RD_THR) ? 1 : 0;
      if (lane == 0) flag[wave] = any_over;
      __syncthreads();
      int f = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) f |= flag[k];
      if (__builtin_amdgcn_readfirstlane(f)) continue;
    }

    // ---- statistics: the last slot's sum, row sums of the two key halves
    {
      float v = lcur + __shfl_xor(lcur, 16);
      v += __shfl_xor(v, 32);
      const float lall = l + lcur;
      float lt = lall + __shfl_xor(lall, 16);
      lt += __shfl_xor(lt, 32);
      if (lb == 0) {
        if (sum_t >= 0) sl_sum[(sum_t * 2 + kh) * 64 + qg * 16 + jq] = v;
        l_ex[kh * 64 + qg * 16 + jq] = lt;
        if (kh == 0) mx_ex[qg * 16 + jq] = m;          // (reference of query qg*16 + jq, for the writers below)
      }
    }
    __syncthreads();
    if (tid < 64) {
      float* mlp = a.ml + ((long)z * a.Npad + qtile * 64 + tid) * 2;
      const float mq = mx_ex[tid];
      mlp[0] = mq > -2.9e38f ? mq * LN2 : RD_NEG;
      mlp[1] = l_ex[tid] + l_ex[64 + tid];
    }
    if (lslot_out)
      for (int e = tid; e < 64 * a.T; e += 512) {
        const int qq = e & 63, t = e >> 6;
        float* lsp = lslot_out + (((long)z * a.Npad + qtile * 64 + qq) * a.T + t) * 2;
        lsp[0] = sl_sum[(t * 2) * 64 + qq] + sl_sum[(t * 2 + 1) * 64 + qq];
        lsp[1] = mx_ex[qq] * LN2;
      }
    __syncthreads();                                  // the images are dead, the statistics read: LDS is reused below

    // ---- flush O: transposed through LDS (the accumulator layout gives one column per lane, i.e. 4-byte
    // stores; staged as [32 rows][128 columns] per wave the tile leaves as 16-byte stores of 512-byte rows)
    {
      char* stg = smem + wave * 16384;
      int lrow = hi, lcol = j;                        // opaque: keeps the row pointers out of the prologue
      R6_OPAQUE(lrow);
      R6_OPAQUE(lcol);
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lrow;
#pragma unroll
          for (int ci = 0; ci < 4; ++ci)
            *reinterpret_cast<float*>(stg + row * 512 + (ci * 32 + lcol) * 4) = o[qi][ci][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave: LDS operations complete in order
        float* orow = a.part + ((long)z * a.Npad + qtile * 64 + qi * 32) * a.ncols + wave * 128;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int row = it * 2 + lrow;
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(stg + row * 512 + lcol * 16);
          *reinterpret_cast<f32x4_t*>(orow + (long)row * a.ncols + lcol * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next round overwrites
      }
    }
    break;
  }
  if (trace && lane == 0) {
    if (wave == 0) {
      trace[3] = __builtin_readcyclecounter();
      trace[28] = n;
    }
    trace[4 + wave] = tacc[0];
    trace[12 + wave] = tacc[1];
    trace[20 + wave] = tacc[2];
    trace[40 + wave] = tacc[3];
    trace[32 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave / SIMD / CU this wave ran on
  }
}

// mode 0 (bank: long-term / self) and mode 1 (window) are separate instantiations: the window arithmetic's state
// (query coordinates, bias row pointer) would otherwise sit in registers of the bank read, which has none to spare
template <int TRACE, int VAR = 0>
__device__ __forceinline__ void read64_body(const rmem_read_args& a, const int blk, char* smem, long long* trace_base = nullptr) {
  if (a.mode == 0) read64_mode<TRACE, VAR, 0>(a, blk, smem, trace_base);
  else read64_mode<TRACE, VAR, 1>(a, blk, smem, trace_base);
}

__global__ __launch_bounds__(512) void read64_kernel(rmem_read_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  read64_body<0>(a, blockIdx.x, smem);
}

template <int VAR>
__global__ __launch_bounds__(512) void read64_trace_kernel(rmem_read_args a, long long* trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  read64_body<1, VAR>(a, blockIdx.x, smem, trace);
}

// The bank read (p[0], mode 0) and the windowed read (p[1], mode 1) of one layer in ONE launch.  Per
// XCD (block % 8) the first cha blocks serve p[0]'s units, the next chb blocks p[1]'s.
struct Read2Args {
  rmem_read_args p[2];
  int cha, chb;
};

__global__ __launch_bounds__(512) void read64x2_kernel(Read2Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int which = jj < g.cha ? 0 : 1;
  read64_body<0>(g.p[which], (which ? jj - g.cha : jj) * 8 + xcd, smem);
}

// the same two kernels for several clips in one launch (launch.h): block z = clip, whose argument
// block is read from device memory
__global__ __launch_bounds__(512) void read64_many_kernel(const char* __restrict__ argv, long stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const rmem_read_args a = rmem::uniform_copy(reinterpret_cast<const rmem_read_args*>(argv + (long)blockIdx.z * stride));
  read64_body<0>(a, blockIdx.x, smem);
}

__global__ __launch_bounds__(512) void read64x2_many_kernel(const char* __restrict__ argv, long stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Read2Args& g = *reinterpret_cast<const Read2Args*>(argv + (long)blockIdx.z * stride);
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int which = jj < __builtin_amdgcn_readfirstlane(g.cha) ? 0 : 1;
  const rmem_read_args a = rmem::uniform_copy(&g.p[which]);
  read64_body<0>(a, (which ? jj - g.cha : jj) * 8 + xcd, smem);
}

template <class K>
static int read_many_thunk(K kernel, const rmem::RecOp& op, const char* dev_args, long stride, int B, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
  hipLaunchKernelGGL(kernel, dim3(op.grid.x, 1, B), dim3(512), R6_LDS, s, dev_args + op.off, stride);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}
static int read_many(const rmem::RecOp& op, const char* d, long st, int B, hipStream_t s) {
  return read_many_thunk(&read64_many_kernel, op, d, st, B, s);
}
static int read2_many(const rmem::RecOp& op, const char* d, long st, int B, hipStream_t s) {
  return read_many_thunk(&read64x2_many_kernel, op, d, st, B, s);
}

static int read_args_ok(const rmem_read_args& a) {
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) != 0 || a.T <= 0 || a.T > 16 || a.ksplits <= 0 || a.ksplits > 32) return 0;
  if (!a.qh || !a.ql || !a.kh || !a.kl || !a.vh || !a.vl || !a.part || !a.ml) return 0;
  if (a.ncols != 1024) return 0;                      // eight waves x 128 columns of [V | ID_V]
  if (a.mode == 1 && (!a.R || a.h * a.w != a.N || a.T != 1 || a.w < 1 || a.ldr < 1)) return 0;
  if (a.mode != 0 && a.mode != 1) return 0;
  return 1;
}

static int read_chunk(const rmem_read_args& a) {
  return (((a.N + 63) / 64) * a.ksplits + 7) / 8;
}

extern "C" int rmem_attn_read2(const rmem_read_args* ap, const rmem_read_args* bp, void* stream) {
  if (!ap || !bp || !read_args_ok(*ap) || !read_args_ok(*bp) || ap->mode != 0 || bp->mode != 1) return RMEM_ERR_INVALID;
  const int cha = read_chunk(*ap), chb = read_chunk(*bp);
  // per launch: the attribute belongs to the (device, function) pair; no process-wide "already set" flag
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read64x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
  Read2Args g;
  g.p[0] = *ap;
  g.p[1] = *bp;
  g.cha = cha;
  g.chb = chb;
  if (rmem::Recorder* r = rmem::current_recorder()) {
    rmem::rec_push(r, &read2_many, dim3(8 * (cha + chb)), dim3(512), R6_LDS, &g, (unsigned)sizeof(g));
    return RMEM_OK;
  }
  hipLaunchKernelGGL(read64x2_kernel, dim3(8 * (cha + chb)), dim3(512), R6_LDS, static_cast<hipStream_t>(stream), g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_attn_read(const rmem_read_args* ap, void* stream) {
  if (!ap || !read_args_ok(*ap)) return RMEM_ERR_INVALID;
  const rmem_read_args& a = *ap;
  const int chunk = read_chunk(a);
  if (rmem::Recorder* r = rmem::current_recorder()) {
    rmem::rec_push(r, &read_many, dim3(8 * chunk), dim3(512), R6_LDS, &a, (unsigned)sizeof(a));
    return RMEM_OK;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
  hipLaunchKernelGGL(read64_kernel, dim3(8 * chunk), dim3(512), R6_LDS, static_cast<hipStream_t>(stream), a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// Debug aid (tools/kbench_read.py): the same launch with shader-clock stamps per block written to
// trace[block][64]: [0] start, [1] reference pass done, [2] tile loop done, [3] end, [4 + w] / [12 + w] / [20 + w] cycles
// wave w spent in its SCORE / PV phases / waiting at the interval barriers, [28] key tiles of the unit, [32 + w]
// HW_REG_HW_ID of wave w, [40 + w] cycles at the top of the iterations (K request, first V requests).  trace must
// hold 8 * ceil(units / 8) * 64 int64.
extern "C" int rmem_attn_read_trace(const rmem_read_args* ap, int64_t* trace, void* stream) {
  if (!ap || !trace || !read_args_ok(*ap) || rmem::current_recorder()) return RMEM_ERR_INVALID;
  const int chunk = read_chunk(*ap);
  const char* ev = getenv("RMEM_READ_VAR");           // experiments (see read64_body)
  const int var = ev ? atoi(ev) & 8 : 0;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
    hipLaunchKernelGGL(kern, dim3(8 * chunk), dim3(512), R6_LDS, static_cast<hipStream_t>(stream), *ap,
                       reinterpret_cast<long long*>(trace));
  };
  if (var == 0) go(&read64_trace_kernel<0>);
  else go(&read64_trace_kernel<8>);
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
