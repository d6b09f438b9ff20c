// linear_stream.h -- the streaming projection kernel (included by linear.hip after linear_epilogue / GroupedLinear).  This is
// its second form (round 6); the first (round 4, `git log -- rmem_amd/csrc/linear_stream_v1.h`) measured as follows.
//
// What the first form measured, per workgroup, on the grouped front launch of a GPM layer
// (rmem_linear_trace, profiles/r06a_stream_trace.json; 35.8 k cycles for two items of four stages = 6.1 k cycles of MFMA):
//   3.9 k cycles from launch to the first request: ten DEPENDENT scalar-load round trips -- the per-problem descriptor was
//         read field by field through dynamic indexes into the kernel argument, each field where it was first used;
//   1.9 k in the first stage of every item: the same chain again for the next item's descriptor, inside the loop;
//   3.2-7.9 k per item epilogue: argument fields loaded one at a time in the middle of the stores, sixteen 4-byte (or
//         2-byte) stores per lane with 64-bit address arithmetic each, per-row and per-plane branches around every one.
// The MFMA stream itself is untouched here: same stages, same ring, same order of operations per output element
// (gemm_mainloop's: k ascending, per 16-deep step hi.lo, lo.hi, hi.hi) -- results equal the tile kernels' bit for bit.
//
// What changed:
//   * ONE round trip per descriptor.  Per problem a 256-byte StreamDesc, 64-byte aligned in the kernel argument, read with
//     s_load_dwordx16 through the constant address space from the kernarg segment pointer: the load side takes three
//     loads issued together, the epilogue two.  Everything the kernel needs of a problem is in it, precomputed by the host
//     (reciprocals, tile counts, destination kind, vector-store eligibility).
//   * Rolled epilogue through LDS.  The accumulators (+ bias, SiLU) leave in four rounds of eight rows: four ds_write_b32
//     at (e * 64 + lane) * 4 -- linear, conflict-free -- and ONE ds_read_b128 at lane * 16 give every lane four
//     consecutive columns of one row; fp32 destinations and split-K partials then take ONE 16-byte store per round and
//     lane (4 per wave instead of 16 four-byte ones), row-major planes two 8-byte stores instead of eight 2-byte ones.  The
//     loop is rolled (the accumulator vector is rotated by four registers per round): ~60 instructions fetched once
//     instead of ~300 -- every launch starts with a cold instruction cache.  Blocked-16 planes (V operand layout) keep the
//     accumulator's own layout -- four consecutive rows of a column ARE its 8-byte unit -- in the same rolled form.
//     Blocked-16 tiles that cross M (one row tile in 27) take element stores under row checks; shapes the fast forms do not
//     cover (per-row bias, accumulate, two fp32 destinations) are served by the tile-per-workgroup kernels.
//   * 8 KB of LDS for the staging (1 KB per wave) behind the ring: 159 KB per workgroup.
#pragma once

// One persistent workgroup per CU (8 waves, two per SIMD) walks its share of the launch's items (problem, 64-row tile,
// 128-column tile, K range) as ONE stream of k-tile stages: operands by LDS-DMA (global_load_lds_dwordx4: no staging
// registers) into a ring of three 48 KB stages requested ACROSS item boundaries; a stage = X tile [64][64 k] + Y tile
// [128][64 k], hi / lo planes, each row 128 B with its 16-byte chunks XOR-swizzled (the DMA lands lane-linear, so the swizzle
// is applied to the SOURCE address); the bias of the item's columns rides along as a seventh request per wave and stage.
template <int NS>
struct StreamCfg {
  static constexpr int BM = 64, BN = 128, BK = 64, NSPLIT = NS;
  static constexpr int THREADS = 512;
  static constexpr int WM = 32, WN = 32, TM = 1, TN = 1;
  static constexpr int NPL = (NS == 1) ? 1 : 2;
  static constexpr int X_BYTES = BM * 128, Y_BYTES = BN * 128;       // one plane of a stage
  static constexpr int STAGE_BYTES = NPL * (X_BYTES + Y_BYTES);
  static constexpr int NSTAGE = 3;
  static constexpr int BIAS = NSTAGE * STAGE_BYTES;                   // [stage][wave][64 floats]: bias of the wave's columns
  static constexpr int ITEMS = BIAS + NSTAGE * 8 * 256;               // [4] x int4: the items in flight (load side -> compute side)
  static constexpr int DUMMY = ITEMS + 64;                            // 1 KiB nobody reads: target of requests past the last stage
  static constexpr int LDS_BYTES = DUMMY + 1024;
  static constexpr int DMA_PER_WAVE = 3 * NPL + 1;                    // X rows 8w.., Y rows 8w.. and 64 + 8w.. per plane, + bias
};

typedef int __attribute__((ext_vector_type(16))) i32x16_t;
typedef int __attribute__((ext_vector_type(8))) i32x8_t;
typedef float __attribute__((ext_vector_type(4))) f32x4s_t;
typedef unsigned int __attribute__((ext_vector_type(2))) u32x2s_t;
typedef const char __attribute__((address_space(4))) * kconst_ptr_t;

enum { S2_GENERIC = 0, S2_F32V = 1, S2_F32S = 2, S2_PLANES = 3, S2_BLOCKED = 4 };
enum { S2F_SILU = 1, S2F_BIAS = 2, S2F_PARTS = 4 };

// Per-problem descriptor.  Bytes 0-127: the load side; 128-159: shared; 128-255: the epilogue.
struct alignas(64) StreamDesc {
  const h16_t* x[2][2];        // [K segment][plane]
  const h16_t* y[2][2];
  int ldx[2], ldy[2];          // leading dimensions per segment (elements)
  long bsx, bsy;               // batch strides (elements)
  int ktx_split, kty_split;    // k-tiles served by the first segment (1 << 30: one segment)
  int M, N;
  int mt, nt;                  // 64-row tiles, 128-column tiles
  float inv_mt, inv_mtnt;
  // ---- byte 128
  int kt_total, ksplits, per, kind;
  const float* bias;           // per-column bias (nullptr: none)
  long bsbias;
  // ---- byte 160
  void* dst[4];                // F32*: [0] = destination (split-K: parts); PLANES: pah, pal, pbh, pbl; BLOCKED: pah, pal
  const float* addvec;         // PLANES with a second set
  long bsd;                    // elements between batches (split-K: between splits) of the destination
  int ld0, ld1;                // F32*: row stride, column stride; PLANES: ldpa, ldpb; BLOCKED: ldpa
  int flags, pad0;
  long pad1, pad2;
};
static_assert(sizeof(StreamDesc) == 256, "StreamDesc is four 64-byte scalar loads");

struct StreamGroup2 {
  StreamDesc d[8];             // offset 0 of the kernarg segment
  int tile_start[9];           // (64-byte aligned: one s_load_dwordx8 + one dword)
  int n;
};

// destinations are global memory: stores through address space 1 (the pointers are assembled from descriptor words, which
// would otherwise make them generic -- flat_store, counted on both memory counters)
#define RMEM_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T RMEM_GLOBAL* gptr(T* p) { return (T RMEM_GLOBAL*)p; }

template <class T>
__device__ __forceinline__ T kload(kconst_ptr_t p) {
  return *reinterpret_cast<const T __attribute__((address_space(4)))*>(p);
}

// ---- epilogue rounds.  Before the call: acc[r] = value of row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 of
// the wave's 32 x 32 tile (bias and activation applied).  stg: this wave's KiB of staging.
// Round g (0-3) covers rows 8 g .. 8 g + 7: after the exchange lane L holds row 8 g + 4 ((L >> 3) & 1) + (L >> 4), columns
// 4 (L & 7) .. + 3.
__device__ __forceinline__ f32x4s_t stage_round(const f32x16_t& acc, char* stg, int lane) {
  float* w = reinterpret_cast<float*>(stg) + lane;
  w[0] = acc[0];
  w[64] = acc[1];
  w[128] = acc[2];
  w[192] = acc[3];
  return *reinterpret_cast<const f32x4s_t*>(stg + lane * 16);       // (one wave: LDS operations complete in order)
}
__device__ __forceinline__ void rotate4(f32x16_t& acc) {
  acc = __builtin_shufflevector(acc, acc, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3);
}

// fp32 destination with contiguous, 16-byte aligned columns (N % 4 == 0): one store per round and lane
__device__ __forceinline__ void ep2_f32v(f32x16_t& acc, char* stg, int lane, float* dst, long ld, int row0, int col0, int M, int N) {
  const int rr = 4 * ((lane >> 3) & 1) + (lane >> 4), col = col0 + 4 * (lane & 7);
  float RMEM_GLOBAL* o = gptr(dst) + (long)(row0 + rr) * ld + col;
  const bool cok = col < N;
#pragma clang loop unroll(disable)
  for (int g = 0; g < 4; ++g) {
    const f32x4s_t t = stage_round(acc, stg, lane);
    if (cok && row0 + 8 * g + rr < M) *reinterpret_cast<f32x4s_t RMEM_GLOBAL*>(o) = t;
    o += 8 * ld;
    rotate4(acc);
  }
}

// fp32 destination, any column stride / alignment: four 4-byte stores per round and lane
__device__ __forceinline__ void ep2_f32s(f32x16_t& acc, char* stg, int lane, float* dst, long ld, long cs, int row0, int col0, int M,
                                         int N) {
  const int rr = 4 * ((lane >> 3) & 1) + (lane >> 4), col = col0 + 4 * (lane & 7);
  float RMEM_GLOBAL* o = gptr(dst) + (long)(row0 + rr) * ld + (long)col * cs;
#pragma clang loop unroll(disable)
  for (int g = 0; g < 4; ++g) {
    const f32x4s_t t = stage_round(acc, stg, lane);
    if (row0 + 8 * g + rr < M) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < N) o[e * cs] = t[e];
    }
    o += 8 * ld;
    rotate4(acc);
  }
}

__device__ __forceinline__ void split4(const f32x4s_t t, u32x2s_t& wh, u32x2s_t& wl) {
  h16_t hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split_f16(t[e], hh[e], ll[e]);
  wh[0] = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16);
  wh[1] = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
  wl[0] = (uint32_t)ll[0] | ((uint32_t)ll[1] << 16);
  wl[1] = (uint32_t)ll[2] | ((uint32_t)ll[3] << 16);
}

// row-major hi / lo planes (+ a second pair holding value + addvec[col]); N % 4 == 0, leading dimensions % 4 == 0
__device__ __forceinline__ void ep2_planes(f32x16_t& acc, char* stg, int lane, h16_t* pah, h16_t* pal, h16_t* pbh, h16_t* pbl,
                                           const float* addvec, long lda, long ldb, int row0, int col0, int M, int N) {
  const int rr = 4 * ((lane >> 3) & 1) + (lane >> 4), col = col0 + 4 * (lane & 7);
  const bool cok = col < N;
  long oa = (long)(row0 + rr) * lda + col, ob = (long)(row0 + rr) * ldb + col;
  f32x4s_t av = {0.f, 0.f, 0.f, 0.f};
  if (pbh && addvec && cok) av = *reinterpret_cast<const f32x4s_t RMEM_GLOBAL*>(gptr(addvec) + col);
#pragma clang loop unroll(disable)
  for (int g = 0; g < 4; ++g) {
    const f32x4s_t t = stage_round(acc, stg, lane);
    if (cok && row0 + 8 * g + rr < M) {
      u32x2s_t wh, wl;
      split4(t, wh, wl);
      *reinterpret_cast<u32x2s_t RMEM_GLOBAL*>(gptr(pah) + oa) = wh;
      if (pal) *reinterpret_cast<u32x2s_t RMEM_GLOBAL*>(gptr(pal) + oa) = wl;
      if (pbh) {
        f32x4s_t u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = t[e] + av[e];
        split4(u, wh, wl);
        *reinterpret_cast<u32x2s_t RMEM_GLOBAL*>(gptr(pbh) + ob) = wh;
        if (pbl) *reinterpret_cast<u32x2s_t RMEM_GLOBAL*>(gptr(pbl) + ob) = wl;
      }
    }
    oa += 8 * lda;
    ob += 8 * ldb;
    rotate4(acc);
  }
}

// blocked-16 planes, tile fully inside M: registers 4 g .. 4 g + 3 of a lane are four consecutive rows of its column = one
// 8-byte unit of the layout ((row / 16) * ldpa + col) * 16 + row % 16
__device__ __forceinline__ void ep2_blocked(f32x16_t& acc, int lane, h16_t* pah, h16_t* pal, long ldpa, int row0, int col0, int N) {
  const int col = col0 + (lane & 31);
  if (col >= N) return;
  const int r0 = row0 + 4 * (lane >> 5);                       // rows r0 + 8 g .. + 3: r0 % 4 == 0, inside one 16-row block
  long off = ((long)(r0 >> 4) * ldpa + col) * 16 + (r0 & 15);
#pragma clang loop unroll(disable)
  for (int g = 0; g < 4; ++g) {
    f32x4s_t t;
    t[0] = acc[0];
    t[1] = acc[1];
    t[2] = acc[2];
    t[3] = acc[3];
    u32x2s_t wh, wl;
    split4(t, wh, wl);
    *reinterpret_cast<u32x2s_t RMEM_GLOBAL*>(gptr(pah) + off) = wh;
    if (pal) *reinterpret_cast<u32x2s_t RMEM_GLOBAL*>(gptr(pal) + off) = wl;
    off += (g & 1) ? (ldpa * 16 - 8) : 8;                       // rows + 8: the other half of the block, then the next block
    rotate4(acc);
  }
}

// the same for a tile that crosses M (one row tile of a problem at most): element stores under row checks
__device__ __forceinline__ void ep2_blocked_tail(f32x16_t& acc, int lane, h16_t* pah, h16_t* pal, long ldpa, int row0, int col0, int M,
                                                 int N) {
  const int col = col0 + (lane & 31);
  if (col >= N) return;
  const int r0 = row0 + 4 * (lane >> 5);
#pragma clang loop unroll(disable)
  for (int g = 0; g < 4; ++g) {
    const int r = r0 + 8 * g;
    const long off = ((long)(r >> 4) * ldpa + col) * 16 + (r & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (r + e < M) {
        h16_t hi, lo;
        split_f16(acc[e], hi, lo);
        gptr(pah)[off + e] = hi;
        if (pal) gptr(pal)[off + e] = lo;
      }
    rotate4(acc);
  }
}

template <int NS>
struct Stream2Cfg : StreamCfg<NS> {
  using B = StreamCfg<NS>;
  static constexpr int STAGING = B::LDS_BYTES;                  // [wave][1 KiB]
  static constexpr int LDS_BYTES = B::LDS_BYTES + 8 * 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

// Iteration structure, ring, request order and MFMA order are the first form's (see linear_stream_v1.h); the comments here
// are about what differs.
template <int NS, int TRACE>
__global__ __launch_bounds__(512) void linear_stream2_kernel(StreamGroup2 g, long long* trace) {
  using Cfg = Stream2Cfg<NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* tr = (TRACE && trace) ? trace + (long)blockIdx.x * 64 : nullptr;
  if (TRACE && tr && threadIdx.x == 0) {
    tr[0] = __builtin_readcyclecounter();
    tr[60] = __builtin_amdgcn_s_memrealtime();       // 100 MHz, one counter for the whole device: launch shape across workgroups
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // the descriptors are read from the kernarg segment itself (g is its first member: offset 0)
  const kconst_ptr_t kbase = (kconst_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  int ts[9];
  {
    const i32x8_t t0 = kload<i32x8_t>(kbase + offsetof(StreamGroup2, tile_start));
    ts[0] = t0[0]; ts[1] = t0[1]; ts[2] = t0[2]; ts[3] = t0[3]; ts[4] = t0[4]; ts[5] = t0[5]; ts[6] = t0[6]; ts[7] = t0[7];
    ts[8] = kload<int>(kbase + offsetof(StreamGroup2, tile_start) + 32);
  }
  const int nprob = kload<int>(kbase + offsetof(StreamGroup2, n));
  int total = ts[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) total = i <= nprob ? ts[i] : total;
  const int G = gridDim.x;
  if ((int)blockIdx.x >= total) return;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  int* items = reinterpret_cast<int*>(smem + Cfg::ITEMS);

  int a_off[4], b_off[4];
  {
    const int ar = wr * 32 + (lane & 31), br = wc * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a_off[ks] = lds_swz(ar, ks * 2 + (lane >> 5));
      b_off[ks] = Cfg::NPL * Cfg::X_BYTES + lds_swz(br, ks * 2 + (lane >> 5));
    }
  }
  const int prow = wave * 8 + (lane >> 3);
  const int pchunk = (lane & 7) ^ ((prow >> 1) & 7);

  // ---- load side.  The descriptor of the current item's problem (its first 160 bytes) lives in `q`: three scalar loads
  // issued together, one wait.
  // (unpacked into scalars through readfirstlane: left as vector elements, the selects between two of them -- first or
  // second K segment -- become dynamically indexed extracts of a VGPR copy of the vector, with a waterfall loop)
  int qa[32], qc[8];
  auto q_fetch = [&](int i) __attribute__((always_inline)) {
    const kconst_ptr_t dp = kbase + (long)i * (long)sizeof(StreamDesc);
    i32x16_t a = kload<i32x16_t>(dp), b = kload<i32x16_t>(dp + 64);
    i32x8_t c = kload<i32x8_t>(dp + 128);
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c));                  // all three requested before the first is used
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      qa[w] = __builtin_amdgcn_readfirstlane(a[w]);
      qa[16 + w] = __builtin_amdgcn_readfirstlane(b[w]);
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) qc[w] = __builtin_amdgcn_readfirstlane(c[w]);
  };
  // dword map of StreamDesc: x[s][p] at 4 s + 2 p, y[s][p] at 8 + 4 s + 2 p, ldx[s] 16 + s, ldy[s] 18 + s, bsx 20, bsy 22,
  // ktx_split 24, kty_split 25, M 26, N 27, mt 28, nt 29, inv_mt 30, inv_mtnt 31 | qc: kt_total 0, ksplits 1, per 2, kind 3,
  // bias 4, bsbias 6
  auto q_i = [&](int w) __attribute__((always_inline)) { return qa[w]; };
  auto q_l = [&](int w) __attribute__((always_inline)) {
    return (long)(((unsigned long)(unsigned)qa[w + 1] << 32) | (unsigned)qa[w]);
  };
  auto q_ptr = [&](int w) __attribute__((always_inline)) { return reinterpret_cast<const char*>(q_l(w)); };
  int ld_idx = blockIdx.x, ld_seq = 0, ld_p = 0, ld_kt = 0, ld_kt1 = 0, ld_seg_end = 0, ld_bz = 0, ld_mx = 0, ld_ny = 0;
  bool ld_valid = true;
  const char* xs[2] = {nullptr, nullptr};
  const char* ys[2] = {nullptr, nullptr};
  long xo = 0, yo0 = 0, yo1 = 0;
  const char* bsrc = nullptr;
  auto ld_segment = [&]() __attribute__((always_inline)) {
    const int ktxs = q_i(24), ktys = q_i(25), M = q_i(26), N = q_i(27);
    const int sx = ld_kt >= ktxs ? 1 : 0, sy = ld_kt >= ktys ? 1 : 0;
    const int ktx = ld_kt - (sx ? ktxs : 0), kty = ld_kt - (sy ? ktys : 0);
    const long bx = ((long)ld_bz * q_l(20) + ktx * 64) * 2, by = ((long)ld_bz * q_l(22) + kty * 64) * 2;
#pragma unroll
    for (int pl = 0; pl < Cfg::NPL; ++pl) {
      xs[pl] = (sx ? q_ptr(4 + 2 * pl) : q_ptr(2 * pl)) + bx;
      ys[pl] = (sy ? q_ptr(12 + 2 * pl) : q_ptr(8 + 2 * pl)) + by;
    }
    const int ldxs = sx ? q_i(17) : q_i(16), ldys = sy ? q_i(19) : q_i(18);
    int rx = ld_mx * 64 + prow, ry0 = ld_ny * 128 + prow, ry1 = ry0 + 64;
    rx = rx < M ? rx : M - 1;
    ry0 = ry0 < N ? ry0 : N - 1;
    ry1 = ry1 < N ? ry1 : N - 1;
    xo = ((long)rx * ldxs + pchunk * 8) * 2;
    yo0 = ((long)ry0 * ldys + pchunk * 8) * 2;
    yo1 = ((long)ry1 * ldys + pchunk * 8) * 2;
    int e = ld_kt1;
    if (!sx && ktxs < e) e = ktxs;
    if (!sy && ktys < e) e = ktys;
    ld_seg_end = e;
  };
  auto ld_item = [&]() __attribute__((always_inline)) {
    int i = 0, start = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const bool in = k < nprob && ld_idx >= ts[k];
      i = in ? k : i;
      start = in ? ts[k] : start;
    }
    ld_p = i;
    q_fetch(i);
    const int mt = q_i(28), nt = q_i(29);
    int local = ld_idx - start;
    ld_bz = __builtin_amdgcn_readfirstlane(fast_div(local, __builtin_bit_cast(float, q_i(31))));
    local -= ld_bz * mt * nt;
    ld_ny = __builtin_amdgcn_readfirstlane(fast_div(local, __builtin_bit_cast(float, q_i(30))));
    ld_mx = local - ld_ny * mt;
    const int kt_total = qc[0], ksplits = qc[1], per = qc[2];
    ld_kt = 0;
    ld_kt1 = kt_total;
    if (ksplits > 1) {
      ld_kt = ld_bz * per;
      ld_kt1 = ld_kt + per < kt_total ? ld_kt + per : kt_total;
    }
    if (tid == 0) {
      int4 d;
      d.x = ld_p;
      d.y = ld_mx | (ld_ny << 16);
      d.z = ld_bz;
      d.w = ld_kt1 - ld_kt;
      *reinterpret_cast<int4*>(items + (ld_seq & 3) * 4) = d;
    }
    ++ld_seq;
    const int bzb = ksplits > 1 ? 0 : ld_bz;
    int bcol = ld_ny * 128 + wc * 32 + (lane & 31);
    const int N = q_i(27);
    bcol = bcol < N ? bcol : N - 1;
    const char* bias = reinterpret_cast<const char*>(((unsigned long)(unsigned)qc[5] << 32) | (unsigned)qc[4]);
    const long bsbias = (long)(((unsigned long)(unsigned)qc[7] << 32) | (unsigned)qc[6]);
    // (a problem without a per-column bias: any valid address, the value is not used)
    bsrc = bias ? bias + ((long)bzb * bsbias + bcol) * 4 : q_ptr(0);
    ld_bz = bzb;
    ld_segment();
  };
  auto piece = [&](int buf, auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    if constexpr (j == 0) {
      const unsigned db = __builtin_amdgcn_readfirstlane(lds0 + (ld_valid ? Cfg::BIAS + (buf * 8 + wave) * 256 : Cfg::DUMMY));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(db), "v"(bsrc) : "memory");
    } else {
      constexpr int pl = (j - 1) / 3, w = (j - 1) % 3;
      const int base = ld_valid ? buf * Cfg::STAGE_BYTES + wave * 1024 : Cfg::DUMMY;
      const int on = ld_valid ? 1 : 0;
      constexpr int off = w == 0 ? pl * Cfg::X_BYTES : Cfg::NPL * Cfg::X_BYTES + pl * Cfg::Y_BYTES + (w == 2 ? 8192 : 0);
      const char* gp = w == 0 ? xs[pl] + xo : (w == 1 ? ys[pl] + yo0 : ys[pl] + yo1);
      const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + base + on * off);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(gp) : "memory");
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    if (ld_valid) {
      ++ld_kt;
#pragma unroll
      for (int pl = 0; pl < Cfg::NPL; ++pl) {
        xs[pl] += 128;
        ys[pl] += 128;
      }
      if (ld_kt >= ld_seg_end) {
        if (ld_kt >= ld_kt1) {
          ld_idx += G;
          ld_valid = ld_idx < total;
          if (ld_valid) {
            ld_item();
          } else {
#pragma unroll
            for (int pl = 0; pl < Cfg::NPL; ++pl) {
              xs[pl] -= 128;
              ys[pl] -= 128;
            }
          }
        } else {
          ld_segment();
        }
      }
    }
  };

  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  frag8_t fa[4][Cfg::NPL], fb[4][Cfg::NPL];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int pl = 0; pl < Cfg::NPL; ++pl)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fa[ks][pl][e] = 0;
        fb[ks][pl][e] = 0;
      }
  float bc_cur = 0.f, bc_next = 0.f;

  ld_item();
  if (TRACE && tr && threadIdx.x == 0) tr[1] = __builtin_readcyclecounter();
  int cp_seq = 0, cp_left = 0, nst = 0;
#pragma clang loop unroll(disable)
  for (int it = -3;; ++it) {
    if (it >= -1) {
      if (Cfg::DMA_PER_WAVE == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (TRACE && tr && threadIdx.x == 0 && it >= 0 && it < 30) tr[2 + 2 * it] = __builtin_readcyclecounter();
    const bool compute = it >= 0, refill = it >= -1;
    if (compute && cp_left == 0)
      cp_left = __builtin_amdgcn_readfirstlane(items[(cp_seq & 3) * 4 + 3]);
    bc_cur = bc_next;
    const int nbuf = (it + 3) % Cfg::NSTAGE;
    const int rbuf = (it + 4) % Cfg::NSTAGE;
    const char* st = smem + rbuf * Cfg::STAGE_BYTES;
    static_for<4>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      if (compute && TRACE != 3) {             // (TRACE 2 / 3 / 4: timing experiments of the tracing kernel -- no requests /
        if constexpr (NS == 3) {               //  no MFMAs / no fragment reads; results are wrong)  small terms first (gemm_mainloop's order)
          acc = RMEM_MFMA(fa[ks][0], fb[ks][1], acc);
          acc = RMEM_MFMA(fa[ks][1], fb[ks][0], acc);
        }
        acc = RMEM_MFMA(fa[ks][0], fb[ks][0], acc);
      }
      if (refill && TRACE != 4) {
#pragma unroll
        for (int pl = 0; pl < Cfg::NPL; ++pl) {
          fa[ks][pl] = *reinterpret_cast<const frag8_t*>(st + a_off[ks] + pl * Cfg::X_BYTES);
          fb[ks][pl] = *reinterpret_cast<const frag8_t*>(st + b_off[ks] + pl * Cfg::Y_BYTES);
        }
        if constexpr (ks == 0)
          bc_next = *reinterpret_cast<const float*>(smem + Cfg::BIAS + (rbuf * 8 + wave) * 256 + (lane & 31) * 4);
      }
      constexpr int per = (Cfg::DMA_PER_WAVE + 3) / 4;
      static_for<per>([&](auto I) {
        constexpr int j = ks * per + decltype(I)::value;
        if constexpr (j < Cfg::DMA_PER_WAVE && TRACE != 2) piece(nbuf, std::integral_constant<int, j>{});
      });
    });
    advance();
    if (compute && --cp_left == 0) {           // item done
      const int4 d = *reinterpret_cast<const int4*>(items + (cp_seq & 3) * 4);
      const int p = __builtin_amdgcn_readfirstlane(d.x), mn = __builtin_amdgcn_readfirstlane(d.y);
      const int bzz = __builtin_amdgcn_readfirstlane(d.z);
      const int m0 = (mn & 0xffff) * 64, n0 = (mn >> 16) * 128;
      // the epilogue's half of the descriptor + (M, N): three scalar loads, one wait
      const kconst_ptr_t dp = kbase + (long)p * (long)sizeof(StreamDesc);
      i32x16_t e0 = kload<i32x16_t>(dp + 128), e1 = kload<i32x16_t>(dp + 192);
      int M = kload<int>(dp + 104), N = kload<int>(dp + 108);
      asm volatile("" : "+s"(e0), "+s"(e1), "+s"(M), "+s"(N));
      // dwords of e0: kt_total 0, ksplits 1, per 2, kind 3, bias 4-5, bsbias 6-7, dst[0] 8-9, dst[1] 10-11, dst[2] 12-13,
      // dst[3] 14-15; e1: addvec 0-1, bsd 2-3, ld0 4, ld1 5, flags 6
      auto e_ptr = [&](const i32x16_t& v, int w) __attribute__((always_inline)) {
        return reinterpret_cast<char*>(((unsigned long)(unsigned)v[w + 1] << 32) | (unsigned)v[w]);
      };
      const int kind = e0[3], flags = e1[6];
      const long bsd = (long)(((unsigned long)(unsigned)e1[3] << 32) | (unsigned)e1[2]);
      const bool full = m0 + 64 <= M;          // (uniform)
      if (n0 + wc * 32 < N) {                  // (wave-uniform: a 4-column problem keeps one wave in four)
        const float bc = ((flags & S2F_BIAS) && (!(flags & S2F_PARTS) || bzz == 0)) ? bc_cur : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc[r] + bc;
        if (flags & S2F_SILU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = silu_f(acc[r]);
        }
        const int row0 = m0 + wr * 32, col0 = n0 + wc * 32;
        char* stg = smem + Cfg::STAGING + wave * 1024;
        const long ld0 = e1[4], ld1 = e1[5];
        if (kind == S2_F32V) {
          ep2_f32v(acc, stg, lane, reinterpret_cast<float*>(e_ptr(e0, 8)) + bzz * bsd, ld0, row0, col0, M, N);
        } else if (kind == S2_F32S) {
          ep2_f32s(acc, stg, lane, reinterpret_cast<float*>(e_ptr(e0, 8)) + bzz * bsd, ld0, ld1, row0, col0, M, N);
        } else if (kind == S2_PLANES) {
          h16_t* pal = reinterpret_cast<h16_t*>(e_ptr(e0, 10));
          h16_t* pbh = reinterpret_cast<h16_t*>(e_ptr(e0, 12));
          h16_t* pbl = reinterpret_cast<h16_t*>(e_ptr(e0, 14));
          ep2_planes(acc, stg, lane, reinterpret_cast<h16_t*>(e_ptr(e0, 8)) + bzz * bsd, pal ? pal + bzz * bsd : nullptr,
                     pbh ? pbh + bzz * bsd : nullptr, pbl ? pbl + bzz * bsd : nullptr,
                     reinterpret_cast<const float*>(e_ptr(e1, 0)), ld0, ld1, row0, col0, M, N);
        } else {                               // S2_BLOCKED (the host keeps other shapes away from this kernel)
          h16_t* pah = reinterpret_cast<h16_t*>(e_ptr(e0, 8)) + bzz * bsd;
          h16_t* pal = reinterpret_cast<h16_t*>(e_ptr(e0, 10));
          pal = pal ? pal + bzz * bsd : nullptr;
          if (full) ep2_blocked(acc, lane, pah, pal, ld0, row0, col0, N);
          else ep2_blocked_tail(acc, lane, pah, pal, ld0, row0, col0, M, N);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      ++cp_seq;
      if (TRACE && tr && threadIdx.x == 0 && it < 30) tr[3 + 2 * it] = __builtin_readcyclecounter();
      nst = it + 1;
      if ((int)blockIdx.x + cp_seq * G >= total) break;
    } else if (TRACE && tr && threadIdx.x == 0 && it >= 0 && it < 30) {
      tr[3 + 2 * it] = __builtin_readcyclecounter();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (TRACE && tr && threadIdx.x == 0) {
    tr[62] = nst;
    tr[63] = __builtin_readcyclecounter();
    tr[61] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- host side
static int stream_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}

// The streaming kernel serves a launch issued directly whose problems ask for tile 0 (auto) or 256; tile 64 / 128 / 192
// select the tile-per-workgroup kernels (kept for the recorded launches of several clips -- launch.h -- and as the
// bit-identical cross-check), as does rmem_configure("linear_tiles", 1) for every launch.  Stays with the tile kernels as well: items of a
// single stage (see below), a split-K problem whose last split would be empty (ceil division; the stream counts one stage per k-tile of every item), leading
// dimensions or item counts beyond what the packed descriptor holds.
static bool use_stream(const rmem_linear_args* args, int n) {
  if (rmem_config().linear_tiles || rmem::current_recorder()) return false;
  long items = 0;
  for (int i = 0; i < n; ++i) {
    const rmem_linear_args& a = args[i];
    if (a.tile != 0 && a.tile != 256) return false;
    if (a.ldx >= (1L << 30) || a.ldy >= (1L << 30) || a.ldx2 >= (1L << 30) || a.ldy2 >= (1L << 30)) return false;
    if (a.M > 65535 * 64 || a.N > 32767 * 128) return false;
    // an item of ONE stage: the load side runs up to four stages ahead and would publish the descriptor of item i + 4
    // into the four-entry item ring before the epilogue of item i has read its slot (a workgroup with five or more
    // items); such shapes (K = 64, or split-K down to one k-tile per split) stay with the tile kernels
    if (a.K / 64 < 2) return false;
    if (a.ksplits > 1) {
      const int kt = a.K / 64, per = (kt + a.ksplits - 1) / a.ksplits;
      if ((a.ksplits - 1) * per >= kt) return false;
      if (per < 2 || kt - (a.ksplits - 1) * per < 2) return false;
    }
    items += (long)((a.M + 63) / 64) * ((a.N + 127) / 128) * (a.ksplits > 1 ? a.ksplits : (a.nbatch > 0 ? a.nbatch : 1));
  }
  return items < (1L << 20);                   // (fast_div's exact range)
}

static bool aligned_to(const void* p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

static void stream2_desc(const rmem_linear_args& a, StreamDesc& q) {
  memset(&q, 0, sizeof(q));
  q.x[0][0] = a.xh; q.x[0][1] = a.xl; q.x[1][0] = a.xh2; q.x[1][1] = a.xl2;
  q.y[0][0] = a.yh; q.y[0][1] = a.yl; q.y[1][0] = a.yh2; q.y[1][1] = a.yl2;
  q.ldx[0] = (int)a.ldx; q.ldx[1] = (int)a.ldx2; q.ldy[0] = (int)a.ldy; q.ldy[1] = (int)a.ldy2;
  q.ktx_split = a.xh2 ? a.kx_split / 64 : (1 << 30);
  q.kty_split = a.yh2 ? a.ky_split / 64 : (1 << 30);
  q.bsx = a.bsx; q.bsy = a.bsy;
  q.M = a.M; q.N = a.N;
  q.mt = (a.M + 63) / 64; q.nt = (a.N + 127) / 128;
  q.inv_mt = 1.0f / (float)q.mt;
  q.inv_mtnt = 1.0f / (float)(q.mt * q.nt);
  q.kt_total = a.K / 64;
  q.ksplits = a.ksplits > 1 ? a.ksplits : 1;
  q.per = (q.kt_total + q.ksplits - 1) / q.ksplits;
  q.bias = (a.bias && !a.bias_per_row) ? a.bias : nullptr;
  q.bsbias = a.bsbias;
  q.flags = (a.act == 1 ? S2F_SILU : 0) | (q.bias ? S2F_BIAS : 0) | (a.ksplits > 1 ? S2F_PARTS : 0);
  q.kind = S2_GENERIC;
  const bool plain_act = a.act == 0 || a.act == 1;
  if (a.bias_per_row || !plain_act) return;
  if (a.ksplits > 1) {                          // raw partials [split][M][N] (validate_linear: no activation, no batches)
    q.dst[0] = a.parts;
    q.bsd = a.part_stride;
    q.ld0 = a.N;
    q.ld1 = 1;
    q.kind = ((a.N % 4) == 0 && (a.part_stride % 4) == 0 && aligned_to(a.parts, 16)) ? S2_F32V : S2_F32S;
    return;
  }
  if (a.pa_blocked) {                           // (validate_linear: the only output)
    if (a.ldpa >= (1L << 27) || (a.bspa % 4) != 0 || !aligned_to(a.pah, 8) || (a.pal && !aligned_to(a.pal, 8))) return;
    q.dst[0] = a.pah; q.dst[1] = a.pal;
    q.bsd = a.bspa;
    q.ld0 = (int)a.ldpa;
    q.kind = S2_BLOCKED;
    return;
  }
  if (a.d0 && !a.d1 && !a.pah && !a.pbh && !a.accumulate && a.csplit >= a.N && a.ldd0 < (1L << 31)) {
    const long cs = a.d0_cs > 0 ? a.d0_cs : 1;
    q.dst[0] = a.d0;
    q.bsd = a.bsd;
    q.ld0 = (int)a.ldd0;
    q.ld1 = (int)cs;
    q.kind = (cs == 1 && (a.N % 4) == 0 && (a.ldd0 % 4) == 0 && (a.bsd % 4) == 0 && aligned_to(a.d0, 16)) ? S2_F32V : S2_F32S;
    return;
  }
  if (a.pah && !a.d0 && !a.d1 && a.act == 0 && (a.N % 4) == 0 && (a.ldpa % 4) == 0 && (a.bspa % 4) == 0 && a.ldpa < (1L << 31) &&
      aligned_to(a.pah, 8) && (!a.pal || aligned_to(a.pal, 8))) {
    if (a.pbh && ((a.ldpb % 4) != 0 || a.ldpb >= (1L << 31) || !aligned_to(a.pbh, 8) || (a.pbl && !aligned_to(a.pbl, 8)) ||
                  (a.addvec && !aligned_to(a.addvec, 16))))
      return;
    q.dst[0] = a.pah; q.dst[1] = a.pal; q.dst[2] = a.pbh; q.dst[3] = a.pbl;
    q.addvec = a.addvec;
    q.bsd = a.bspa;
    q.ld0 = (int)a.ldpa;
    q.ld1 = (int)a.ldpb;
    q.kind = S2_PLANES;
  }
}

static int stream2_group(const rmem_linear_args* args, int n, StreamGroup2& g) {
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    stream2_desc(args[i], g.d[i]);
    g.tile_start[i] = total;
    total += g.d[i].mt * g.d[i].nt * (args[i].ksplits > 1 ? args[i].ksplits : (args[i].nbatch > 0 ? args[i].nbatch : 1));
  }
  for (int i = n; i < 8; ++i) memset(&g.d[i], 0, sizeof(StreamDesc));
  for (int i = n; i < 9; ++i) g.tile_start[i] = total;
  return total;
}
// (shapes whose descriptor says S2_GENERIC -- per-row bias, accumulate, two fp32 destinations, planes beside fp32,
// unaligned planes -- are not served by this kernel)
static bool stream2_covers(const StreamGroup2& g) {
  for (int i = 0; i < g.n; ++i)
    if (g.d[i].kind == S2_GENERIC) return false;
  return true;
}

template <int NS, int TRACE>
static int launch_stream2(const StreamGroup2& g, int total, long long* trace, hipStream_t s) {
  using Cfg = Stream2Cfg<NS>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_stream2_kernel<NS, TRACE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            Cfg::LDS_BYTES);
  const int grid = total < stream_cus() ? total : stream_cus();
  hipLaunchKernelGGL((linear_stream2_kernel<NS, TRACE>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, g, trace);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}
