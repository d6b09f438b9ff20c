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
// Per 64-key tile there are two kinds of work that touch disjoint buffers:
//   SCORE(i+1): S^T = K.Q^T for 16 queries x ALL 64 keys on v_mfma_f32_16x16x32_f16, K and Q fragments from LDS;
//       "swapped" (rows = keys) so that a query's scores live in 4 lanes; weights P = 2^(y - m) go to the other P
//       image as fp16 hi / lo planes.  Wave w scores query group w & 3 on the tiles whose parity is w >> 2: a
//       query's whole row of a tile is in ONE wave, so the running reference m of the row is a wave-local decision;
//   PV(i): O += P.V for all 64 queries x this wave's 128 columns on v_mfma_f32_32x32x16_f16: A = P fragments from
//       LDS (shared by the eight waves), B = V fragments straight from global memory ("blocked-16" layout: one
//       contiguous KiB per load instruction).
// One barrier per tile, and SCORE in two parts so that every wave has a P.V and half a SCORE per interval: in interval
// i the four waves that own tile i + 1 finish SCORE(i+1) (part 2: exp2, fp16 planes -> P) and then run PV(i); the other
// four run PV(i) and then start SCORE(i+2) (part 1: the 48 small MFMAs, logits, row maxima; the logits wait in
// registers across the barrier).  Wave w and wave w + 4 share a SIMD, so on each SIMD one wave feeds the matrix pipe
// while the other reads LDS, takes exp2 and converts.  K tiles arrive by LDS-DMA (global_load_lds_dwordx4, three
// tiles ahead, chunk swizzle applied to the SOURCE address: no staging registers, no ds_write phase); P and K are
// double-buffered.
//
// Online reference, lazily raised.  m of a row starts at the maximum of its first tile and is raised to a tile's
// maximum only when that exceeds m by more than RD_BUMP = 12 (log2 domain): weights stay <= 2^12 (fp16 hi plane),
// and the accumulators are rescaled only on those tiles -- the owner publishes the factor 2^(m_old - m_new) per row and
// a flag in LDS, and every wave multiplies its O rows at the start of that tile's P.V (a wave-uniform branch that real
// data takes on the first tiles of a unit and when a dominant key arrives).  No reference pass, no second attempt.
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
constexpr int R6_SL = 131072;               // per-slot sums [16 slots][2 tile parities][64 queries] fp32 ...
constexpr int R6_SM = R6_SL + 8192;         // ... and the reference each was taken against
constexpr int R6_MX = R6_SM + 8192;         // [64 queries] current reference m of the row (log2 domain)
constexpr int R6_RS = R6_MX + 512;          // [4 tiles in flight][64] factor 2^(m_old - m_new) of a tile that raised m
constexpr int R6_L = R6_RS + 1024;           // [2][64] row sums
constexpr int R6_FL = R6_L + 512;           // [4 tiles in flight][4 query groups] "this tile raised m"
constexpr int R6_DUMMY = R6_FL + 64;        // 1 KiB nobody reads: target of the K requests that fetch no tile
constexpr int R6_LDS = R6_DUMMY + 1024;
constexpr float RD_NEG = -3.0e38f;
constexpr float RD_BUMP = 12.0f;            // log2 domain: weights up to 2^12 = 4096 << 65504 (fp16 hi plane)

#define R6_OPAQUE(x) asm volatile("" : "+v"(x))

// VAR (experiments, tracing kernel only): 4 = no s_setprio at all; 8 = never wait for V fragments (timing only:
// results are wrong); 16 = one MFMA per product instead of three (hi . hi only; every plane is still staged: what the
// exactness of split-fp16 costs in matrix-pipe time, profiles/r04_read_one_product_probe.md).
// zwin: the key splits this launch class serves -- z0 = zwin & 0xffff first split, nz = zwin >> 16 how many (0 = all of them)
template <int TRACE, int VAR, int MODE>
__device__ __forceinline__ void read64_mode(const rmem_read_args& a, const int blk, char* smem, long long* trace_base, const int zwin) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qg = wave & 3, grp = wave >> 2;           // score role: query group (16 queries), parity of the tiles it scores
  const int jq = lane & 15, lb = lane >> 4;           // 16x16 tile: column (query) / row block (4 keys)
  const int j = lane & 31, hi = lane >> 5;            // 32x32 tile (P.V)

  // ---- work unit.  Units are ordered (split, query tile); every XCD (block b runs on XCD b % 8:
  // observed placement, used for speed only) owns a contiguous chunk, so the units of one XCD share a
  // key split, i.e. the same K / V bytes in that XCD's L2.
  const int nq = (a.N + 63) / 64;
  const int z0 = zwin & 0xffff, nz = (zwin >> 16) ? (zwin >> 16) : a.ksplits;
  const int nunits = nq * nz;
  const int chunk = (nunits + 7) / 8;
  const int jj = blk >> 3;
  const int u = (blk & 7) * chunk + jj;
  if (jj >= chunk || u >= nunits) return;
  const int qtile = u % nq;
  const int z = z0 + u / nq;

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
  // Even split, or (mode 0, a.nfull > 0) UNEVEN: the first nfull splits hold a.pf key tiles each, the remaining ones share
  // the rest evenly.  The short pieces are launched LAST (rmem_attn_read2) and run behind the windowed units of the same
  // launch on the CUs those leave early: 27 x (7 x 14 + 2 x 5) tiles instead of 27 x (6 x 16 + 12) at 480p K=4.
  int lo, hi_t;
  if (MODE == 0 && a.nfull > 0) {
    if (z < a.nfull) {
      lo = k_lo + z * a.pf;
      hi_t = lo + a.pf < k_hi ? lo + a.pf : k_hi;
    } else {
      const int base = k_lo + a.nfull * a.pf, nr = a.ksplits - a.nfull;
      const int pr = (k_hi - base + nr - 1) / nr;
      lo = base + (z - a.nfull) * pr;
      hi_t = lo + pr < k_hi ? lo + pr : k_hi;
    }
  } else {
    const int per = (k_hi - k_lo + a.ksplits - 1) / a.ksplits;
    lo = k_lo + z * per;
    hi_t = lo + per < k_hi ? lo + per : k_hi;
  }
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
  if (trace && tid == 0) {
    trace[0] = __builtin_readcyclecounter();
    trace[48] = __builtin_amdgcn_s_memrealtime();     // 100 MHz, one counter for the whole device: launch shape across blocks
  }

  SlotLut lut;
  lut.load(a.slot_map, MODE == 0 ? a.T : 1);

  float* sl_sum = reinterpret_cast<float*>(smem + R6_SL);
  float* sl_m = reinterpret_cast<float*>(smem + R6_SM);
  float* mrow = reinterpret_cast<float*>(smem + R6_MX);
  float* resc = reinterpret_cast<float*>(smem + R6_RS);
  float* l_ex = reinterpret_cast<float*>(smem + R6_L);
  int* bflag = reinterpret_cast<int*>(smem + R6_FL);

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
  int ak0 = R6_K + jq * 64 + ((lb ^ fsw) << 4);                       // K row kt*16 + jq;           + k4*4096 + kt*1024
  int aq0 = R6_Q + (qg * 16 + jq) * 64 + ((lb ^ fsw) << 4);           // Q row qg*16 + jq;           + k4*4096
  R6_OPAQUE(ak0);
  R6_OPAQUE(aq0);
  auto ak = [&](int k4) __attribute__((always_inline)) { return ak0 + k4 * 4096; };
  auto aq = [&](int k4) __attribute__((always_inline)) { return aq0 + k4 * 4096; };
  int apw0;                                           // P stores: row qg*16 + jq, keys kt*16 + lb*4 .. +3 = k-step kt
  {
    const int row = qg * 16 + jq;
    apw0 = R6_P + row * 32 + (((lb >> 1) ^ ((row >> 3) & 1)) << 4) + (lb & 1) * 8;
    R6_OPAQUE(apw0);
  }
  auto apw = [&](int kt) __attribute__((always_inline)) { return apw0 + kt * 2048; };
  int apr0 = R6_P + j * 32 + ((hi ^ ((j >> 3) & 1)) << 4);            // P fragments: row j (+ 32 qi), chunk hi; + ks*2048 + qi*1024
  R6_OPAQUE(apr0);
  auto apr = [&](int ks) __attribute__((always_inline)) { return apr0 + ks * 2048; };
  // Staging (LDS-DMA): a wave-instruction moves 1 KiB = 16 rows x 64 B of
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
  // Windowed read: relative bias of this lane's 8 keys key0 + half*32 + kt*16 + lb*4 + e (kt = 0, 1) of the tile at
  // key0 -> rb[half*8 ..], and a bit in vm for each key inside the 15x15 window and the image.  All gathers are issued
  // unconditionally (index 0 where masked) and back to back; one division per 4 consecutive keys.
  auto window_terms = [&](int key0, int half, float* rb, unsigned& vm) __attribute__((always_inline)) {
    int idx[8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int tok0 = key0 + half * 32 + kt * 16 + lb * 4;
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
        idx[kt * 4 + e] = valid ? ((dy + 7) * 15 + dx + 7) * rcs : 0;
        vm |= valid ? 1u << (half * 8 + kt * 4 + e) : 0u;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) rb[half * 8 + r] = Rq[idx[r]];
  };

  // ---- scores of one tile for this lane: y[kt*4 + r] = log2-domain logit of key kt*16 + lb*4 + r, kt = 0..3 (RD_NEG
  // where masked), the three-product form.  kb = byte offset of the tile's K image relative to R6_K.
  int cur_t = -1;
  float bias2 = 0.f;
  auto scores64 = [&](const TileIter& ti, int kb, float (&y)[16]) __attribute__((always_inline)) {
    const int t = ti.t, key0 = ti.key0();
    if (MODE == 0 && t != cur_t) {
      cur_t = t;
      bias2 = ((a.bias && qvalid) ? a.bias[(long)q * a.T + t] : 0.f) * sl2e;     // (padding queries: no row in bias)
    }
    f32x4_t s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] = 0.f;
    // 16 items = (d-step k4, key group kt); the fragments of item i + 1 are requested before the MFMAs of item i (two
    // register sets of 2 fragments, + two of Q): read one item at a time the phase is sixteen LDS round trips long
    frag8_t fq[2][2], fk[2][2];                       // [set][hi, lo]: Q of a d-step; K of a (d-step, key group)
    auto qload = [&](frag8_t (&f)[2], auto K4) __attribute__((always_inline)) {
      constexpr int k4 = decltype(K4)::value;
      f[0] = *reinterpret_cast<const frag8_t*>(smem + aq(k4));
      f[1] = *reinterpret_cast<const frag8_t*>(smem + aq(k4) + 16384);
    };
    auto kload = [&](frag8_t (&f)[2], auto K4, auto KT) __attribute__((always_inline)) {
      constexpr int k4 = decltype(K4)::value, kt = decltype(KT)::value;
      f[0] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + kt * 1024);
      f[1] = *reinterpret_cast<const frag8_t*>(smem + ak(k4) + kb + (16384 + kt * 1024));
    };
    qload(fq[0], std::integral_constant<int, 0>{});
    kload(fk[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    static_for<16>([&](auto I) {
      constexpr int i = I.value, k4 = i >> 2, kt = i & 3;
      if constexpr (i + 1 < 16) {
        constexpr int n4 = (i + 1 < 16 ? (i + 1) >> 2 : 0), nkt = (i + 1) & 3;
        if constexpr (nkt == 0) qload(fq[n4 & 1], std::integral_constant<int, n4>{});
        kload(fk[(i + 1) & 1], std::integral_constant<int, n4>{}, std::integral_constant<int, nkt>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      frag8_t (&fqc)[2] = fq[k4 & 1];
      frag8_t (&fkc)[2] = fk[i & 1];
      if constexpr (!(VAR & 16)) {                    // (VAR 16, timing only: the hi . hi product alone, operands still loaded)
        s[kt] = RMEM_MFMA16(fkc[0], fqc[1], s[kt]);   // small terms first: K hi . Q lo, K lo . Q hi, K hi . Q hi
        s[kt] = RMEM_MFMA16(fkc[1], fqc[0], s[kt]);
      }
      s[kt] = RMEM_MFMA16(fkc[0], fqc[0], s[kt]);
      __builtin_amdgcn_sched_barrier(0);
    });
    const bool padded = key0 + 64 > a.N;
    if (MODE == 0 && !padded) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[kt * 4 + r] = fmaf(s[kt][r], sl2e, bias2);
    } else if (MODE == 0) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = key0 + kt * 16 + lb * 4 + r;
          y[kt * 4 + r] = tok < a.N ? fmaf(s[kt][r], sl2e, bias2) : RD_NEG;
        }
    } else {
      // (the 16 gathers issued BEFORE the score MFMAs, to land behind them, was built: their 17 registers push address
      // registers of this phase to scratch, and each reload waits for everything in flight -- 58 against 56 us)
      float rbw[16];
      unsigned vm = 0;
      window_terms(key0, 0, rbw, vm);
      window_terms(key0, 1, rbw, vm);
#pragma unroll
      for (int e = 0; e < 16; ++e)
        y[e] = ((vm >> e) & 1u) ? fmaf(s[e >> 2][e & 3], sl2e, rbw[e] * 1.44269504088896341f) : RD_NEG;
    }
    if constexpr (TRACE == 2) {
      // debug (rmem_attn_read_trace with a.dbg_logits): every pre-softmax logit this lane computed, in the natural-log
      // domain the reference's QK tensor is in (attention.py:184, :344) -- keys outside the image / window are left alone
      if (a.dbg_logits && qvalid) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int tok = key0 + (e >> 2) * 16 + lb * 4 + (e & 3);
          if (tok < a.N && y[e] > 0.5f * RD_NEG)
            a.dbg_logits[(long)q * a.dbg_ld + (long)(MODE == 0 ? t : 0) * a.N + tok] = y[e] * 0.693147180559945f;
        }
      }
    }
    return t;
  };

  int vcol = (wave * 128 + j) * 16 + hi * 8;           // V fragments: lane = column, 8 consecutive keys (16 B)
  R6_OPAQUE(vcol);

  // ---- statistics start empty; first K tiles
  for (int e = tid; e < 2048; e += 512) {
    sl_sum[e] = 0.f;
    sl_m[e] = RD_NEG;
  }
  if (tid < 64) mrow[tid] = RD_NEG;
  if (tid < 16) bflag[tid] = 0;
  {
    TileIter t01;
    tinit(t01, lo);
    dma_k(t01, 0, true);
    tstep(t01);
    if (n > 1) dma_k(t01, 1, true);
  }
  // Q and K(0) are in (requests retire in order: at most the 4 pieces of K(1) may still be in flight -- its first
  // reader runs behind the next barrier)
  if (n > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (trace && tid == 0) trace[1] = __builtin_readcyclecounter();

  // ================= the pass
  f32x16_t o[2][4];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qi][ci][r] = 0.f;
  float l = 0.f, lcur = 0.f;                          // this lane's part of the row sum (all slots / the current slot) ...
  float m_known = RD_NEG;                             // ... taken against this reference
  int sum_t = -1;                                     // slot lcur belongs to
  const int row_s = qg * 16 + jq;                     // this lane's query in the score role
  u32x4_t vr[4][2];                                   // ring of V fragments: step s = (k-step, ci) lives in vr[s & 3][plane]

  // re-express the sums against the row's reference `mc` (a wave-uniform branch; taken when the other parity's owner, or
  // this wave, raised m)
  auto follow = [&](float mc) __attribute__((always_inline)) {
    if (__any(mc != m_known)) {
      const float f = __builtin_amdgcn_exp2f(m_known - mc);       // (first reference: 2^(-3e38 - mc) = 0, the sums are 0 too)
      l *= f;
      lcur *= f;
      m_known = mc;
    }
  };

  // SCORE(i) by its owner, in two parts that run in consecutive intervals (see the loop):
  //   part 1 -- scores of tile i (K image kbuf), the row maxima, the decision to raise m (slot i & 3 of the flags /
  //             factors: the P.V of tile i reads it two intervals later);
  //   part 2 -- weights against the reference part 1 left, fp16 hi / lo planes -> P image pbuf, row sums.
  // Between the two the logits and the reference stay in registers (across ONE barrier, no P.V in between).
  float yc[16];
  float mcc = RD_NEG;
  int raised = 0;                                     // bit s: this wave set flag slot s (s = tile & 3, two of them are its own)
  auto score_p1 = [&](const TileIter& ti, int x, int kbuf) __attribute__((always_inline)) {
    const int fs = x & 3;
    if (raised & (1 << fs)) {                         // (wave-uniform) the P.V that read this slot is four barriers back
      if (lane == 0) bflag[fs * 4 + qg] = 0;
      raised &= ~(1 << fs);
    }
    float mc = mrow[row_s];
    scores64(ti, kbuf * 32768, yc);
    follow(mc);
    // the row's largest score of this tile (sentinels are far below): 4 lanes hold a query
    float rm = fmaxf(fmaxf(fmaxf(fmaxf(yc[0], yc[1]), fmaxf(yc[2], yc[3])), fmaxf(fmaxf(yc[4], yc[5]), fmaxf(yc[6], yc[7]))),
                     fmaxf(fmaxf(fmaxf(yc[8], yc[9]), fmaxf(yc[10], yc[11])), fmaxf(fmaxf(yc[12], yc[13]), fmaxf(yc[14], yc[15]))));
    {                                                   // max over lanes ^ 16, ^ 32: lane swaps, no ds_bpermute round trips
      float plo, phi;
      xor16_pair(rm, plo, phi);
      rm = fmaxf(plo, phi);
      xor32_pair(rm, plo, phi);
      rm = fmaxf(plo, phi);
    }
    const bool bump = rm > mc + RD_BUMP;              // (mc = -3e38 before the first valid key: any valid score raises it)
    if (__any(bump)) {                                // (wave-uniform, rare) raise m of those rows to this tile's maximum
      const float mn = bump ? rm : mc;
      if (lb == 0) {
        resc[fs * 64 + row_s] = __builtin_amdgcn_exp2f(mc - mn);    // 1 for the rows that stay
        mrow[row_s] = mn;
      }
      if (lane == 0) bflag[fs * 4 + qg] = 1;
      raised |= 1 << fs;
      mc = mn;
      follow(mc);
    }
    mcc = mc;
  };
  auto score_p2 = [&](const TileIter& ti, int pbuf) __attribute__((always_inline)) {
    const int t = ti.t;
    if (t != sum_t) {
      if (sum_t >= 0) {                               // (wave-uniform) slot finished: park its sum and what it refers to
        float plo, phi;                                 // lcur + lcur[^16], then + [^32] (the pair forms: a + b = b + a)
        xor16_pair(lcur, plo, phi);
        float v = plo + phi;
        xor32_pair(v, plo, phi);
        v = plo + phi;
        if (lb == 0) {
          sl_sum[(sum_t * 2 + grp) * 64 + row_s] = v;
          sl_m[(sum_t * 2 + grp) * 64 + row_s] = m_known;
        }
      }
      lcur = 0.f;
      sum_t = t;
    }
    const float mc = mcc;
    const bool masked = MODE == 1 || ti.key0() + 64 > a.N;      // (wave-uniform) the tile may hold RD_NEG sentinels
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x2_t pp[2];
      if (!masked) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(yc[kt * 4 + e] - mc);
          psum += p;
          pp[e >> 1][e & 1] = p;
        }
      } else {
        asm volatile("" ::: "memory");                // keeps this form a branch (as a select it costs every tile two more VALU per key)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = yc[kt * 4 + e];
          float p = __builtin_amdgcn_exp2f(sv - mc);  // sentinels (-3e38) give exactly 0 unless m is one too
          p = sv > -2.9e38f ? p : 0.f;
          psum += p;
          pp[e >> 1][e & 1] = p;
        }
      }
      // hi = fp16(p), lo = fp16(p - hi), round to nearest even: one packed conversion and two mixed-precision fmas per pair
      uint32_t h0, l0, h1, l1;
      split_pair_f16(pp[0][0], pp[0][1], h0, l0);
      split_pair_f16(pp[1][0], pp[1][1], h1, l1);
      u32x2_t wh, wl;
      wh[0] = h0;
      wh[1] = h1;
      wl[0] = l0;
      wl[1] = l1;
      *reinterpret_cast<u32x2_t*>(smem + apw(kt) + pbuf * 16384) = wh;
      *reinterpret_cast<u32x2_t*>(smem + apw(kt) + pbuf * 16384 + 8192) = wl;
    }
    l += psum;
    lcur += psum;
  };

  // ---- V fragments: a ring of 4 steps (step = k-step x 32-column tile, 8 registers) that runs THROUGH the tile
  // boundaries.  Requested by inline assembly four steps ahead and awaited with COUNTED vmcnt tied to the ring
  // registers ("+v"): left to the compiler, its wait-count pass protects the ring with s_waitcnt vmcnt(0), i.e. waits
  // for the K transfer that was just requested.  The compiler sees no vector-memory load in the P.V cluster.
  const h16_t* vhp = a.vh;                            // blocked-16 planes of the tile whose P.V runs / of the next tile
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
    u32x4_t& d0 = vr[sidx & 3][0];                    // (asm operands do not capture: name the slots first)
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
  // whose load has not landed.)  Loads the compiler issues itself in SCORE (a bias on a slot change, the window
  // gathers) are younger than all of these and waited for before SCORE ends: by then nothing is in flight.
  auto vwait = [&](auto S) __attribute__((always_inline)) {
    constexpr int sidx = decltype(S)::value;
    u32x4_t& d0 = vr[sidx & 3][0];
    u32x4_t& d1 = vr[sidx & 3][1];
    if constexpr (VAR & 8) {                          // (timing experiment of the tracing kernel: never wait; wrong results)
      asm volatile("" : "+v"(d0), "+v"(d1));
    } else if constexpr (sidx < 4) {
      asm volatile("s_waitcnt vmcnt(10)" : "+v"(d0), "+v"(d1));
    } else {
      asm volatile("s_waitcnt vmcnt(6)" : "+v"(d0), "+v"(d1));
    }
  };
  // PV(i): O += P(pbuf) . V(tile i).  16 steps (k-step, 32-column tile) of 6 MFMAs; the V fragments of a step are
  // requested 4 steps ahead (the last four steps request the first four of the next tile), the P fragments of a
  // k-step one k-step ahead.  A tile that raised m: the rows' factors first.
  auto pv_phase = [&](int pbuf, int fs) __attribute__((always_inline)) {
    {
      const int* fp = bflag + fs * 4;
      const int f0 = __builtin_amdgcn_readfirstlane(fp[0]), f1 = __builtin_amdgcn_readfirstlane(fp[1]);
      const int f2 = __builtin_amdgcn_readfirstlane(fp[2]), f3 = __builtin_amdgcn_readfirstlane(fp[3]);
      if (f0 | f1 | f2 | f3) {                        // (wave-uniform, rare)
        // accumulator register r of tile qi is row qi*32 + (r & 3) + 8 (r >> 2) + 4 hi: registers 0-7 belong to query
        // group 2 qi, 8-15 to 2 qi + 1
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int fl = qi == 0 ? (hf == 0 ? f0 : f1) : (hf == 0 ? f2 : f3);
            if (fl) {
#pragma unroll
              for (int r8 = 0; r8 < 8; ++r8) {
                const int r = hf * 8 + r8;
                const float f = resc[fs * 64 + qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) o[qi][ci][r] *= f;
              }
            }
          }
      }
    }
    frag8_t pa[4], pb[4];                             // P fragments [qi][plane] of one k-step
    auto pload = [&](frag8_t (&pf)[4], auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          pf[qi * 2 + p] = *reinterpret_cast<const frag8_t*>(smem + apr(ks) + (pbuf * 16384 + p * 8192 + qi * 1024));
    };
    pload(pa, std::integral_constant<int, 0>{});
    static_for<16>([&](auto S) {
      constexpr int sidx = S.value;
      constexpr int ks = sidx >> 2, ci = sidx & 3;
      frag8_t (&pc)[4] = (ks & 1) ? pb : pa;
      if constexpr (ci == 0 && ks < 3) pload((ks & 1) ? pa : pb, std::integral_constant<int, (ks < 3 ? ks + 1 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      vwait(S);
      const frag8_t vh = __builtin_bit_cast(frag8_t, vr[sidx & 3][0]);
      const frag8_t vl = __builtin_bit_cast(frag8_t, vr[sidx & 3][1]);
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {                // small terms first
        if constexpr (!(VAR & 16)) {
          o[qi][ci] = RMEM_MFMA(pc[qi * 2 + 0], vl, o[qi][ci]);
          o[qi][ci] = RMEM_MFMA(pc[qi * 2 + 1], vh, o[qi][ci]);
        }
        o[qi][ci] = RMEM_MFMA(pc[qi * 2 + 0], vh, o[qi][ci]);
      }
      vreq(std::integral_constant<int, sidx + 4>{});  // (steps 16-19: the next tile's first four, see vwait)
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // Interval `it` (ONE barrier ends it): the waves whose parity is that of it + 1 finish SCORE(it + 1) (part 2) and then
  // run PV(it); the others run PV(it) and then start SCORE(it + 2) (part 1).  Every wave has a P.V and half a SCORE per
  // interval, and on each SIMD one wave is in its matrix-heavy phase while the other reads LDS, takes exp2, converts.
  //   K(x) sits in buffer x & 1: read by part 1 of SCORE(x) in interval x - 2, requested at the top of interval x - 3
  //        (its buffer's previous tile was read in interval x - 4), awaited before barrier x - 3;
  //   P(x) in buffer x & 1: written in interval x - 1, read by PV(x); PV(x - 2), the previous reader, ended at barrier x - 2;
  //   flag / factor slot x & 3: written in interval x - 2, read by PV(x).
  TileIter t_p1, t_p2, t_dma, t_v;                    // tiles of part 1 (it + 2), part 2 (it + 1), K request (it + 3), V base (it + 1)
  tinit(t_p1, lo);
  tinit(t_p2, lo);
  tinit(t_dma, lo + 2 < hi_t ? lo + 2 : lo);
  tinit(t_v, lo);
  // prologue: SCORE(0) by parity 0; then K(2) requested, and part 1 of SCORE(1) by parity 1 beside part 2 of SCORE(0)
  if (grp == 0) score_p1(t_p1, 0, 0);
  tstep(t_p1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (K(1))
  __syncthreads();                                    // m after tile 0 visible; K(0) read; K(1) in
  if (n > 2) {
    dma_k(t_dma, 0, true);
    if (lo + 3 < hi_t) tstep(t_dma);
  }
  if (grp == 0) score_p2(t_p2, 0);
  tstep(t_p2);
  if (n > 1) {
    if (grp == 1) score_p1(t_p1, 1, 1);
    tstep(t_p1);
  }
  {
    const long vb = t_v.vslot + (long)(t_v.key0() >> 4) * (1024 * 16);
    vhp = a.vh + vb;
    vlp = a.vl + vb;
    if (n > 1) tstep(t_v);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // K(2)
  vreq(std::integral_constant<int, 0>{});
  vreq(std::integral_constant<int, 1>{});
  vreq(std::integral_constant<int, 2>{});
  vreq(std::integral_constant<int, 3>{});
  __syncthreads();
#pragma clang loop unroll(disable)
  for (int it = 0; it < n; ++it) {
    const bool more = it + 1 < n;
    long long t0 = 0;
    if (TRACE) t0 = __builtin_readcyclecounter();
    // K(it + 3) into the buffer of K(it + 1); when there is no such tile the 4 pieces go to a 1 KiB dummy target (keeps
    // the request count of vwait() constant)
    {
      const bool real = it + 3 < n;
      const long base = t_dma.kslot + (long)t_dma.key0() * 128;
      const int dst = real ? R6_K + ((it + 1) & 1) * 32768 + dma_dst0 : R6_DUMMY;
      const int dstep = real ? 8192 : 0;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + dst + pl * (real ? 16384 : 0) + pc * dstep);
          const char* g = reinterpret_cast<const char*>((pl ? a.kl : a.kh) + base) + dma_off0 + pc * 128;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(g) : "memory");
        }
      if (it + 4 < n) tstep(t_dma);                    // (never past the unit's last tile: the dummy requests re-read that one)
    }
    if (more) {                                       // V planes of the next tile (its first steps are requested in PV)
      const long vb = t_v.vslot + (long)(t_v.key0() >> 4) * (1024 * 16);
      vhn = a.vh + vb;
      vln = a.vl + vb;
      if (it + 2 < n) tstep(t_v);
    }
    if (TRACE) { const long long t1 = __builtin_readcyclecounter(); tacc[3] += t1 - t0; t0 = t1; }
    // Priorities.  (a) A wave that is NOT in its P.V cluster goes first (3): at equal priority the partner's queued
    // MFMAs sit at the head of the SIMD's vector issue and this wave's LDS reads / exp2 / conversions crawl beside
    // them.  (b) When the two P.V clusters of a SIMD collide, the wave that still has its SCORE part AFTER the cluster
    // (P.V -> part 1) must win (1 against 0): it then scores beside the other wave's P.V.  Left to the hardware the
    // older wave (0-3) wins every collision, and in every other interval the loser ends up with P.V and part 1 back
    // to back while the winner waits at the barrier (measured: waves 4-7 6.4 k cycles per P.V against 3.5 k, waves
    // 0-3 3.1 k per tile at the barrier).
    const bool second = ((it + 1) & 1) == grp;        // this wave owns tile it + 1: part 2 now, P.V after
    if (second && more) {
      if (!(VAR & 4)) __builtin_amdgcn_s_setprio(3);
      score_p2(t_p2, (it + 1) & 1);
    }
    if (!(VAR & 4)) {
      if (second) __builtin_amdgcn_s_setprio(0);
      else __builtin_amdgcn_s_setprio(1);
    }
    // (the logits are dead from here to part 1: said explicitly, or they keep 17 registers through the P.V cluster --
    // the compiler cannot see that a wave alternates between the two kinds of interval)
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("" : "=v"(yc[e]));
    asm volatile("" : "=v"(mcc));
    if (TRACE) { const long long t1 = __builtin_readcyclecounter(); tacc[0] += t1 - t0; t0 = t1; }
    pv_phase(it & 1, it & 3);
    if (TRACE) { const long long t1 = __builtin_readcyclecounter(); tacc[1] += t1 - t0; t0 = t1; }
    if (!second && it + 2 < n) {
      if (!(VAR & 4)) __builtin_amdgcn_s_setprio(3);
      score_p1(t_p1, it + 2, it & 1);
    }
    if (!(VAR & 4)) __builtin_amdgcn_s_setprio(0);
    if (more) tstep(t_p2);
    if (it + 2 < n) tstep(t_p1);
    if (more) {
      vhp = vhn;
      vlp = vln;
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // all but the next tile's first four V steps: the K transfer is in
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (TRACE) { const long long t1 = __builtin_readcyclecounter(); tacc[0] += t1 - t0; t0 = t1; }
    __syncthreads();                                  // P(it + 1), K(it + 3), m visible; P(it), K(it + 2) may be overwritten
    if (TRACE) tacc[2] += __builtin_readcyclecounter() - t0;
  }
  if (trace && tid == 0) trace[2] = __builtin_readcyclecounter();

  // ---- statistics against the final reference of the row: the last slot's sum, row sums of the two parities
  {
    const float mfin = mrow[row_s];
    follow(mfin);
    float plo, phi;
    xor16_pair(lcur, plo, phi);
    float v = plo + phi;
    xor32_pair(v, plo, phi);
    v = plo + phi;
    xor16_pair(l, plo, phi);
    float lt = plo + phi;
    xor32_pair(lt, plo, phi);
    lt = plo + phi;
    if (lb == 0) {
      if (sum_t >= 0) {
        sl_sum[(sum_t * 2 + grp) * 64 + row_s] = v;
        sl_m[(sum_t * 2 + grp) * 64 + row_s] = mfin;
      }
      l_ex[grp * 64 + row_s] = lt;
    }
  }
  __syncthreads();
  if (tid < 64) {
    float* mlp = a.ml + ((long)z * a.Npad + qtile * 64 + tid) * 2;
    const float mq = mrow[tid];
    mlp[0] = mq > -2.9e38f ? mq * LN2 : RD_NEG;
    mlp[1] = l_ex[tid] + l_ex[64 + tid];
  }
  if (lslot_out)
    for (int e = tid; e < 64 * a.T; e += 512) {
      const int qq = e & 63, t = e >> 6;
      float* lsp = lslot_out + (((long)z * a.Npad + qtile * 64 + qq) * a.T + t) * 2;
      const float mq = mrow[qq];
      const int i0 = (t * 2) * 64 + qq, i1 = (t * 2 + 1) * 64 + qq;
      // (a parity that saw no tile of the slot holds sum 0 against -3e38: 0 * 2^0 or 0 * 0)
      lsp[0] = sl_sum[i0] * __builtin_amdgcn_exp2f(sl_m[i0] - mq) + sl_sum[i1] * __builtin_amdgcn_exp2f(sl_m[i1] - mq);
      lsp[1] = mq * LN2;
    }
  __syncthreads();                                    // the images are dead, the statistics read: LDS is reused below

  // ---- flush O: transposed through LDS (the accumulator layout gives one column per lane, i.e. 4-byte
  // stores; staged as [32 rows][128 columns] per wave the tile leaves as 16-byte stores of 512-byte rows).
  // One split and a gate given (a.gout): the rows leave normalised and gated -- rmem_attn_read_combine's arithmetic for
  // one split, where the split weight is exp(0) = 1 and the sums collapse: g = (o * (1 / l)) * u -- and no partial is written.
  {
    const bool fused = a.gout != nullptr && a.ksplits == 1;     // (uniform)
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
      if (!fused) {
        float* orow = a.part + ((long)z * a.Npad + qtile * 64 + qi * 32) * a.ncols + wave * 128;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int row = it * 2 + lrow;
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(stg + row * 512 + lcol * 16);
          *reinterpret_cast<f32x4_t*>(orow + (long)row * a.ncols + lcol * 4) = v;
        }
      } else {
        const int q0 = qtile * 64 + qi * 32;
        const float* urow = a.gate + (long)q0 * a.ldgate + wave * 128 + lcol * 4;
        float* grow = a.gout + (long)q0 * a.ldgout + wave * 128 + lcol * 4;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int row = it * 2 + lrow;
          if (q0 + row < a.N) {
            const float inv_l = 1.0f / (l_ex[qi * 32 + row] + l_ex[64 + qi * 32 + row]);
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(stg + row * 512 + lcol * 16);
            const f32x4_t u = *reinterpret_cast<const f32x4_t*>(urow + (long)row * a.ldgate);
            f32x4_t g;
            g[0] = v[0] * inv_l * u[0];
            g[1] = v[1] * inv_l * u[1];
            g[2] = v[2] * inv_l * u[2];
            g[3] = v[3] * inv_l * u[3];
            *reinterpret_cast<f32x4_t*>(grow + (long)row * a.ldgout) = g;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next round overwrites
    }
  }
  if (trace && lane == 0) {
    if (wave == 0) {
      trace[3] = __builtin_readcyclecounter();
      trace[49] = __builtin_amdgcn_s_memrealtime();
      trace[28] = n;
    }
    trace[4 + wave] = tacc[0];
    trace[12 + wave] = tacc[1];
    trace[20 + wave] = tacc[2];
    trace[40 + wave] = tacc[3];
    trace[32 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave / SIMD / CU this wave ran on
  }
}

// mode 0 (bank: long-term / self) and mode 1 (window) are separate instantiations: the window arithmetic's state (query
// coordinates, bias row pointer, the gathered biases) would otherwise hold registers of the bank read, which has none
// to spare, and the bank read's (bias row, slot bookkeeping) registers of the windowed one
template <int TRACE, int VAR = 0>
__device__ __forceinline__ void read64_body(const rmem_read_args& a, const int blk, char* smem, long long* trace_base = nullptr,
                                            const int zwin = 0) {
  if (a.mode == 0) read64_mode<TRACE, VAR, 0>(a, blk, smem, trace_base, zwin);
  else read64_mode<TRACE, VAR, 1>(a, blk, smem, trace_base, 0);
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
// the tracing kernel + the dump of every pre-softmax logit (rmem_read_args.dbg_logits): an instantiation of its own so that
// the stores do not sit in the kernel whose cycle stamps are read as timings
__global__ __launch_bounds__(512) void read64_logits_kernel(rmem_read_args a, long long* trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  read64_body<2, 0>(a, blockIdx.x, smem, trace);
}

// The bank read (p[0], mode 0) and the windowed read (p[1], mode 1) of one layer in ONE launch.  Per
// XCD (block % 8) the first cha blocks serve p[0]'s units, the next chb blocks p[1]'s.
// With uneven long-term splits (p[0].nfull > 0) a third class follows: the SHORT long-term pieces (chs chunks), so that
// the dispatch / queue order is long pieces, windowed units, short pieces -- longest first.
struct Read2Args {
  rmem_read_args p[2];
  int cha, chb, chs;
};

// block index of the paired launch (x = index inside the launch, already XCD-major: xcd = x & 7) -> (argument block, unit
// index inside its class, split window)
__device__ __forceinline__ void read2_class(const Read2Args& g, int x, int& which, int& blk, int& zwin) {
  const int xcd = x & 7, jj = x >> 3;
  const int nfull = g.p[0].nfull;
  if (jj < g.cha) {
    which = 0;
    blk = jj * 8 + xcd;
    zwin = nfull > 0 ? (nfull << 16) : 0;
  } else if (jj < g.cha + g.chb) {
    which = 1;
    blk = (jj - g.cha) * 8 + xcd;
    zwin = 0;
  } else {
    which = 0;
    blk = (jj - g.cha - g.chb) * 8 + xcd;
    zwin = ((g.p[0].ksplits - nfull) << 16) | nfull;
  }
}

__global__ __launch_bounds__(512) void read64x2_kernel(Read2Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int which, blk, zwin;
  read2_class(g, blockIdx.x, which, blk, zwin);
  read64_body<0>(g.p[which], blk, smem, nullptr, zwin);
}

// the same two kernels for several clips in one launch (launch.h): block z = clip, whose argument
// block is read from device memory
__global__ __launch_bounds__(512) void read64_many_kernel(const char* __restrict__ argv, long stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const rmem_read_args a = rmem::uniform_copy(reinterpret_cast<const rmem_read_args*>(argv + (long)blockIdx.z * stride));
  read64_body<0>(a, blockIdx.x, smem);
}

// Block order of the paired read for several clips: ALL clips' long-term units first, then all windowed units (one
// 1-D grid).  A long-term unit is 2-4 times a windowed one and there are more units than CUs (8 clips: 648 on 256):
// clip-major order (every clip's windowed units in the middle of the queue) ended with a tail of long-term units on
// a few CUs -- 725 us for 8 clips against 520 of evenly spread work; longest-first lets the windowed units fill in.
// Block b still runs on XCD b % 8 = the XCD its unit was meant for (both chunk counts are multiples of 8).
__global__ __launch_bounds__(512) void read64x2_many_kernel(const char* __restrict__ argv, long stride, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Read2Args& g0 = *reinterpret_cast<const Read2Args*>(argv);      // (clips of a batch share geometry and splits)
  const int cha = __builtin_amdgcn_readfirstlane(g0.cha), chb = __builtin_amdgcn_readfirstlane(g0.chb);
  const int L = 8 * cha, W = 8 * chb;
  int b = blockIdx.x, clip, x;
  if (b < B * L) {
    clip = b / L;
    x = b - clip * L;
  } else {
    b -= B * L;
    clip = b / W;
    x = L + (b - clip * W);
  }
  const Read2Args& g = *reinterpret_cast<const Read2Args*>(argv + (long)clip * stride);
  const int xcd = x & 7, jj = x >> 3;
  const int which = jj < cha ? 0 : 1;
  const rmem_read_args a = rmem::uniform_copy(&g.p[which]);
  read64_body<0>(a, (which ? jj - cha : jj) * 8 + xcd, smem);
}

// ---- more units than CUs: one workgroup per CU, units taken from a counter, longest first.
// Left to the hardware, block b of a grid larger than the device waits for a free CU of XCD b % 8 in index order:
// 720p K=8 puts 30 long-term units and 8 windowed ones on each XCD's 32 CUs, and the windowed units queue on the
// two CUs that are left (the makespan: 683 us for 630 us of long-term work); 8 clips per launch end with a tail of
// long-term units on a few CUs.  Here workgroup w runs unit w, then unit ncu + (fetch-and-add of sched[0]), ... in
// the launch's unit order (long-term first), whatever XCD it sits on.  sched[1] counts the workgroups that are done;
// the last one zeroes both for the next launch.
__device__ __forceinline__ int next_unit(int* sched, int base, char* smem) {
  int* slot = reinterpret_cast<int*>(smem + R6_DUMMY);          // (any word nobody holds across units)
  __syncthreads();                                     // every wave is done with the unit (LDS, its flush)
  if (threadIdx.x == 0) *slot = base + atomicAdd(sched, 1);
  __syncthreads();
  const int u = *slot;
  __syncthreads();                                     // read before the next unit's first request lands there
  return __builtin_amdgcn_readfirstlane(u);
}
__device__ __forceinline__ void sched_done(int* sched) {
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(sched + 1, 1) == (int)gridDim.x - 1) {
      sched[0] = 0;
      sched[1] = 0;
      __threadfence();
    }
  }
}

__global__ void zero_sched_kernel(int* sched) {
  if (threadIdx.x < 2) sched[threadIdx.x] = 0;
}

__global__ __launch_bounds__(512) void read64x2_pull_kernel(Read2Args g, int* sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = 8 * (g.cha + g.chb + g.chs);
  for (int b = blockIdx.x; b < total; b = next_unit(sched, gridDim.x, smem)) {
    int which, blk, zwin;
    read2_class(g, b, which, blk, zwin);
    read64_body<0>(g.p[which], blk, smem, nullptr, zwin);
  }
  sched_done(sched);
}

__global__ __launch_bounds__(512) void read64x2_many_pull_kernel(const char* __restrict__ argv, long stride, int B, int* sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Read2Args& g0 = *reinterpret_cast<const Read2Args*>(argv);
  const int cha = __builtin_amdgcn_readfirstlane(g0.cha), chb = __builtin_amdgcn_readfirstlane(g0.chb);
  const int L = 8 * cha, W = 8 * chb, total = B * (L + W);
  for (int u = blockIdx.x; u < total; u = next_unit(sched, gridDim.x, smem)) {
    int b = u, clip, x;
    if (b < B * L) {
      clip = b / L;
      x = b - clip * L;
    } else {
      b -= B * L;
      clip = b / W;
      x = L + (b - clip * W);
    }
    const Read2Args& g = *reinterpret_cast<const Read2Args*>(argv + (long)clip * stride);
    const int xcd = x & 7, jj = x >> 3;
    const int which = jj < cha ? 0 : 1;
    const rmem_read_args a = rmem::uniform_copy(&g.p[which]);
    read64_body<0>(a, (which ? jj - cha : jj) * 8 + xcd, smem);
  }
  sched_done(sched);
}

static int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
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
  const int ncu = device_cus();
  int* sched = static_cast<int*>(op.aux);              // (of the first clip's recording; one launch serves all clips)
  if (sched && (int)op.grid.x * B > ncu) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read64x2_many_pull_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
    hipLaunchKernelGGL(zero_sched_kernel, dim3(1), dim3(64), 0, s, sched);   // (a kernel, not a memset node: see rmem_attn_read2)
    hipLaunchKernelGGL(read64x2_many_pull_kernel, dim3(ncu), dim3(512), R6_LDS, s, d + op.off, st, B, sched);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read64x2_many_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
    hipLaunchKernelGGL(read64x2_many_kernel, dim3(op.grid.x * B), dim3(512), R6_LDS, s, d + op.off, st, B);
  }
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

static int read_args_ok(const rmem_read_args& a) {
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) != 0 || a.T <= 0 || a.T > 16 || a.ksplits <= 0 || a.ksplits > 32) return 0;
  if (!a.qh || !a.ql || !a.kh || !a.kl || !a.vh || !a.vl || !a.ml) return 0;
  const bool fused = a.gout != nullptr && a.ksplits == 1;      // (the gated aggregate leaves the read itself: no partial buffer needed)
  if (!fused && !a.part) return 0;
  if (fused && (!a.gate || a.lslot || (a.ldgate % 4) || (a.ldgout % 4) || a.ldgate < a.ncols || a.ldgout < a.ncols)) return 0;
  if (a.ncols != 1024) return 0;                      // eight waves x 128 columns of [V | ID_V]
  if (a.mode == 1 && (!a.R || a.h * a.w != a.N || a.T != 1 || a.w < 1 || a.ldr < 1)) return 0;
  if (a.mode != 0 && a.mode != 1) return 0;
  if (a.nfull < 0 || a.pf < 0) return 0;
  if (a.nfull > 0) {        // uneven splits: mode 0 only, at least one short split, the full pieces must leave keys for the rest
    const int tiles = a.T * ((a.N + 63) / 64);
    if (a.mode != 0 || a.nfull >= a.ksplits || a.pf <= 0 || a.nfull * a.pf >= tiles) return 0;
  }
  return 1;
}

static int read_chunk(const rmem_read_args& a) {
  return (((a.N + 63) / 64) * a.ksplits + 7) / 8;
}

extern "C" int rmem_attn_read2(const rmem_read_args* ap, const rmem_read_args* bp, void* stream) {
  if (!ap || !bp || !read_args_ok(*ap) || !read_args_ok(*bp) || ap->mode != 0 || bp->mode != 1) return RMEM_ERR_INVALID;
  const int nq = (ap->N + 63) / 64;
  const bool uneven = ap->nfull > 0;
  if (uneven && rmem::current_recorder()) return RMEM_ERR_INVALID;      // (several clips per launch: even splits only)
  const int cha = uneven ? (nq * ap->nfull + 7) / 8 : read_chunk(*ap), chb = read_chunk(*bp);
  const int chs = uneven ? (nq * (ap->ksplits - ap->nfull) + 7) / 8 : 0;
  // per launch: the attribute belongs to the (device, function) pair; no process-wide "already set" flag
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read64x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
  Read2Args g;
  g.p[0] = *ap;
  g.p[1] = *bp;
  g.cha = cha;
  g.chb = chb;
  g.chs = chs;
  if (rmem::Recorder* r = rmem::current_recorder()) {
    rmem::rec_push(r, &read2_many, dim3(8 * (cha + chb)), dim3(512), R6_LDS, &g, (unsigned)sizeof(g));
    r->ops.back().aux = ap->sched;
    return RMEM_OK;
  }
  const int ncu = device_cus();
  if (ap->sched && 8 * (cha + chb + chs) > ncu) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&read64x2_pull_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
    // The queue counters are zeroed on the launch stream in front of every pull launch: the last workgroup out also zeroes
    // them, but a faulted or aborted launch would otherwise leave the next read skipping units silently.  By a KERNEL, not
    // hipMemsetAsync: captured into a hipGraph the memset became a memset node, and graphs holding such nodes faulted
    // ("Memory access fault by GPU") when they were replayed after eagerly issued frames had run in between -- the second
    // 720p K=8 clip of a driver, whose first two frames are eager (tools/clip720_twice_probe.py; the plain kernels' graphs,
    // which hold no memset node, never did: profiles/r05_720p_second_clip_fault.md).
    hipLaunchKernelGGL(zero_sched_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), ap->sched);
    hipLaunchKernelGGL(read64x2_pull_kernel, dim3(ncu), dim3(512), R6_LDS, static_cast<hipStream_t>(stream), g, ap->sched);
    RMEM_CHECK_LAUNCH();
    return RMEM_OK;
  }
  hipLaunchKernelGGL(read64x2_kernel, dim3(8 * (cha + chb + chs)), dim3(512), R6_LDS, static_cast<hipStream_t>(stream), g);
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
// trace[block][64]: [0] start, [1] Q / first K tiles in, [2] tile loop done, [3] end, [4 + w] / [12 + w] / [20 + w] cycles
// wave w spent in its SCORE / PV phases / waiting at the interval barriers, [28] key tiles of the unit, [32 + w]
// HW_REG_HW_ID of wave w, [40 + w] cycles at the top of the iterations (K request, first V requests).  trace must
// hold 8 * ceil(units / 8) * 64 int64.
extern "C" int rmem_attn_read_trace(const rmem_read_args* ap, int64_t* trace, void* stream) {
  if (!ap || !trace || !read_args_ok(*ap) || rmem::current_recorder()) return RMEM_ERR_INVALID;
  const int chunk = read_chunk(*ap);
  const int var = rmem_config().read_var;             // experiments (see read64_body)
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R6_LDS);
    hipLaunchKernelGGL(kern, dim3(8 * chunk), dim3(512), R6_LDS, static_cast<hipStream_t>(stream), *ap,
                       reinterpret_cast<long long*>(trace));
  };
  if (ap->dbg_logits) go(&read64_logits_kernel);
  else if (var == 4) go(&read64_trace_kernel<4>);
  else if (var == 8) go(&read64_trace_kernel<8>);
  else if (var == 16) go(&read64_trace_kernel<16>);
  else go(&read64_trace_kernel<0>);
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
    mm = wave_max_xor(mm);
    const float w = lz > 0.f ? expf(mz - mm) : 0.f;
    float L = w * lz;
    L = wave_sum_xor_t(L);
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
