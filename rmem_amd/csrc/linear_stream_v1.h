// linear_stream.h -- the streaming projection kernel (included by linear.hip after linear_epilogue / GroupedLinear).
//
// The projections of the memory path are M = 1674 (480p) x N <= 512 x K <= 2048: a few hundred 64 x 64 tiles of 4-8
// k-tiles each.  Under the tile-per-workgroup kernels of linear.hip a launch costs its fixed parts, not its MFMAs
// (profiles/r04c_kbench_gemm.json, r04e_stream_trace.json): 9.4 us for the 27 tiles of a 4-column problem and 20.4 us for
// the 837 tiles of a layer's grouped front launch, of which the matrix pipe needs 2.4 -- every tile pays a chain of
// dependent scalar argument loads, one memory round trip per k-tile (operands through registers, ds_write_b128, two
// barriers), a bias round trip and a 7 k-cycle generic epilogue.
//
// Here ONE persistent workgroup per CU (8 waves, two per SIMD) walks its share of the launch's work items
// (problem, 64-row tile, 128-column tile, K range) as ONE continuous stream of k-tile stages:
//   * operands by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of three 48 KB
//     stages, requested two stages ahead ACROSS item boundaries -- the first stages of the next item are in flight while
//     this item's last MFMAs and its epilogue run; one raw barrier per stage, counted vmcnt;
//   * a stage = X tile [64 rows][64 k] + Y tile [128 rows][64 k], hi / lo planes, each row 128 B with its 16-byte chunks
//     XOR-swizzled (lds_swz); the DMA lands lane-linear, so the swizzle is applied to the SOURCE address.  The bias of
//     the item's columns rides along as a seventh request per wave and stage (256 B): no bias round trip in the epilogue;
//   * wave (wr, wc) = (w >> 2, w & 3) owns the 32 x 32 accumulator tile at rows 32 wr, columns 32 wc: 16 registers; per
//     stage 16 ds_read_b128 and 12 MFMAs (hi.lo, lo.hi, hi.hi per 16-deep k-step, gemm_mainloop's order: every output
//     element sees the same operations in the same order as under the 4-wave kernels -- bit-identical results);
//   * everything a stage needs from the argument blocks is resolved once per item into registers (packed per-problem
//     descriptor StreamProb, float-reciprocal divisions); the compute side learns the item it is finishing from a
//     four-entry ring in LDS instead of decoding it again;
//   * the epilogue is specialised per destination kind (split-K partials, fp32 [+ SiLU, column stride], blocked-16 planes
//     + SiLU) with the row checks hoisted for full tiles; other shapes take linear_epilogue.
#pragma once

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

enum { SK_GENERIC = 0, SK_PARTS = 1, SK_F32 = 2, SK_F32_SILU = 3, SK_BLOCKED = 4, SK_BLOCKED_SILU = 5, SK_PLANES = 6 };

// What the load side and the item decode need of a problem, packed so that one round of scalar loads fetches it
struct StreamProb {
  const h16_t* x[2][2];        // [K segment][plane]
  const h16_t* y[2][2];
  int ldx[2], ldy[2];          // leading dimensions per segment (elements)
  int ktx_split, kty_split;    // k-tiles served by the first segment (1 << 30: one segment)
  long bsx, bsy;               // batch strides (elements)
  int M, N, mt, nt;            // rows, columns, 64-row tiles, 128-column tiles
  float inv_mt, inv_mtnt;
  int kt_total, ksplits, per, kind;
};

struct StreamGroup {
  int n;
  int tile_start[9];
  StreamProb q[8];
  rmem_linear_args p[8];
};

// ---- specialised epilogues.  acc register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of the
// wave's 32 x 32 tile; bc = bias of the lane's column (0 when the problem has none).
template <bool FULL>
__device__ __forceinline__ void stream_ep_parts(const rmem_linear_args& a, const f32x16_t& acc, int row0, int col, int bzz, float bc) {
  if (col >= a.N) return;
  float* out = a.parts + (long)bzz * a.part_stride + (long)row0 * a.N + col;
  const float b = bzz == 0 ? bc : 0.f;
  const long ld = a.N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dr = (r & 3) + 8 * (r >> 2);
    if (FULL || row0 + dr < a.M) out[dr * ld] = acc[r] + b;
  }
}

template <bool FULL, bool SILU>
__device__ __forceinline__ void stream_ep_f32(const rmem_linear_args& a, const f32x16_t& acc, int row0, int col, int bz, float bc) {
  if (col >= a.N) return;
  float* out = a.d0 + bz * a.bsd + (long)row0 * a.ldd0 + (long)col * (a.d0_cs > 0 ? a.d0_cs : 1);
  const long ld = a.ldd0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dr = (r & 3) + 8 * (r >> 2);
    float v = acc[r] + bc;
    if (SILU) v = silu_f(v);
    if (FULL || row0 + dr < a.M) out[dr * ld] = v;
  }
}

template <bool SILU>
__device__ __forceinline__ void stream_ep_blocked(const rmem_linear_args& a, const f32x16_t& acc, int row0, int col, int bz, float bc) {
  if (col >= a.N) return;
  h16_t* pah = a.pah + bz * a.bspa;
  h16_t* pal = a.pal ? a.pal + bz * a.bspa : nullptr;
#pragma unroll
  for (int g = 0; g < 4; ++g) {                // registers 4g .. 4g + 3: four consecutive rows of one 16-row block
    const int r0 = row0 + 8 * g;
    h16_t hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[4 * g + e] + bc;
      if (SILU) v = silu_f(v);
      split_f16(v, hh[e], ll[e]);
    }
    const long off = ((long)(r0 >> 4) * a.ldpa + col) * 16 + (r0 & 15);
    if (r0 + 3 < a.M) {
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
        if (r0 + e < a.M) {
          pah[off + e] = hh[e];
          if (pal) pal[off + e] = ll[e];
        }
    }
  }
}

template <bool FULL>
__device__ __forceinline__ void stream_ep_planes(const rmem_linear_args& a, const f32x16_t& acc, int row0, int col, int bz, float bc) {
  if (col >= a.N) return;
  h16_t* pah = a.pah + bz * a.bspa + (long)row0 * a.ldpa + col;
  h16_t* pal = a.pal ? a.pal + bz * a.bspa + (long)row0 * a.ldpa + col : nullptr;
  h16_t* pbh = a.pbh ? a.pbh + bz * a.bspa + (long)row0 * a.ldpb + col : nullptr;
  h16_t* pbl = a.pbl ? a.pbl + bz * a.bspa + (long)row0 * a.ldpb + col : nullptr;
  const float addv = (a.pbh && a.addvec) ? a.addvec[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dr = (r & 3) + 8 * (r >> 2);
    if (FULL || row0 + dr < a.M) {
      const float v = acc[r] + bc;
      h16_t hi, lo;
      split_f16(v, hi, lo);
      pah[dr * a.ldpa] = hi;
      if (pal) pal[dr * a.ldpa] = lo;
      if (pbh) {
        split_f16(v + addv, hi, lo);
        pbh[dr * a.ldpb] = hi;
        if (pbl) pbl[dr * a.ldpb] = lo;
      }
    }
  }
}

// Iteration `it` of the stream (it = -3, -2, ... : the first three only fill the pipeline; ONE loop body serves prologue
// and steady state, so that the instructions a workgroup fetches once -- every launch starts with a cold instruction
// cache, ~12 cycles per instruction of straight-line code -- are few):
//   top      (it >= -1)  stage it + 1 has landed: counted wait (the requests of stage it + 2 may stay in flight), barrier --
//                        which also says every wave has the fragments of stage it in registers, so its buffer is free;
//   requests             of stage it + 3 into that buffer, spread behind the MFMA groups (a CU ingests a stage in ~770
//                        cycles at 64 B/clk: issued in one block they hold every wave while the matrix pipe idles);
//   MFMAs    (it >= 0)   of stage it from the fragment registers; behind each k-step's group its registers are refilled
//                        with the fragments of stage it + 1 -- the LDS reads of a stage (128 KB per CU, ~510 cycles) run
//                        under the MFMAs of the stage before instead of in front of their own;
//   item end (it >= 0)   epilogue from the accumulators, the bias (read with the stage's fragments) and the item ring.
template <int NS, int TRACE>
__global__ __launch_bounds__(512) void linear_stream_kernel(StreamGroup g, long long* trace) {
  using Cfg = StreamCfg<NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // TRACE (rmem_linear_trace, tools/kbench_gemm.py): shader-clock stamps of wave 0 of every workgroup, trace[block][64]:
  // [0] start, [1] entering the loop, then per stage s >= 0 [2 + 2 s] top of its iteration passed (stage s + 1 landed),
  // [3 + 2 s] its MFMAs (and, on an item's last stage, the epilogue) issued; [62] stages run, [63] end
  long long* tr = (TRACE && trace) ? trace + (long)blockIdx.x * 64 : nullptr;
  if (TRACE && tr && threadIdx.x == 0) tr[0] = __builtin_readcyclecounter();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int ts[9];                                   // tile_start in registers: the item decode scans it without memory round trips
#pragma unroll
  for (int i = 0; i < 9; ++i) ts[i] = g.tile_start[i];
  const int nprob = g.n;
  int total = ts[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) total = i <= nprob ? ts[i] : total;
  const int G = gridDim.x;
  if ((int)blockIdx.x >= total) return;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  int* items = reinterpret_cast<int*>(smem + Cfg::ITEMS);

  // ---- fragment addresses: row = 32 wr / 32 wc + (lane & 31), 16-byte chunk 2 ks + (lane >> 5) at slot chunk ^ ((row >> 1) & 7)
  int a_off[4], b_off[4];
  {
    const int ar = wr * 32 + (lane & 31), br = wc * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a_off[ks] = lds_swz(ar, ks * 2 + (lane >> 5));
      b_off[ks] = Cfg::NPL * Cfg::X_BYTES + lds_swz(br, ks * 2 + (lane >> 5));
    }
  }
  // ---- DMA source: lane L of a piece (8 rows x 128 B) lands at row L >> 3, slot L & 7 and therefore fetches chunk
  // (L & 7) ^ ((row >> 1) & 7) of its row; the pieces of wave w start at rows 8 w (X, Y) and 64 + 8 w (Y): same parity
  const int prow = wave * 8 + (lane >> 3);
  const int pchunk = (lane & 7) ^ ((prow >> 1) & 7);

  // ---- load side (three stages ahead of the MFMAs).  Uniform: item index / sequence number, problem, k-tile range, the
  // k-tile at which the operand pointers must be re-resolved (second K segment), the pointers at the current k-tile;
  // per lane: byte offsets of the lane's rows (+ chunk) inside the current segment, the bias address.
  int ld_idx = blockIdx.x, ld_seq = 0, ld_p = 0, ld_kt = 0, ld_kt1 = 0, ld_seg_end = 0, ld_bz = 0, ld_mx = 0, ld_ny = 0;
  bool ld_valid = true;
  const char* xs[2] = {nullptr, nullptr};
  const char* ys[2] = {nullptr, nullptr};
  long xo = 0, yo0 = 0, yo1 = 0;
  const char* bsrc = nullptr;
  auto ld_segment = [&]() __attribute__((always_inline)) {      // pointers / offsets for k-tile ld_kt of the current item
    const StreamProb& q = g.q[ld_p];
    const int sx = ld_kt >= q.ktx_split ? 1 : 0, sy = ld_kt >= q.kty_split ? 1 : 0;
    const int ktx = ld_kt - (sx ? q.ktx_split : 0), kty = ld_kt - (sy ? q.kty_split : 0);
    const long bx = (long)ld_bz * q.bsx + ktx * 64, by = (long)ld_bz * q.bsy + kty * 64;
#pragma unroll
    for (int pl = 0; pl < Cfg::NPL; ++pl) {
      xs[pl] = reinterpret_cast<const char*>(q.x[sx][pl] + bx);
      ys[pl] = reinterpret_cast<const char*>(q.y[sy][pl] + by);
    }
    int rx = ld_mx * 64 + prow, ry0 = ld_ny * 128 + prow, ry1 = ry0 + 64;
    rx = rx < q.M ? rx : q.M - 1;
    ry0 = ry0 < q.N ? ry0 : q.N - 1;
    ry1 = ry1 < q.N ? ry1 : q.N - 1;
    xo = ((long)rx * q.ldx[sx] + pchunk * 8) * 2;
    yo0 = ((long)ry0 * q.ldy[sy] + pchunk * 8) * 2;
    yo1 = ((long)ry1 * q.ldy[sy] + pchunk * 8) * 2;
    int e = ld_kt1;                            // next k-tile at which a segment starts, if inside the item
    if (!sx && q.ktx_split < e) e = q.ktx_split;
    if (!sy && q.kty_split < e) e = q.kty_split;
    ld_seg_end = e;
  };
  auto ld_item = [&]() __attribute__((always_inline)) {         // decode item ld_idx, publish it to the compute side
    int i = 0, start = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const bool in = k < nprob && ld_idx >= ts[k];
      i = in ? k : i;
      start = in ? ts[k] : start;
    }
    ld_p = i;
    const StreamProb& q = g.q[i];
    int local = ld_idx - start;
    ld_bz = __builtin_amdgcn_readfirstlane(fast_div(local, q.inv_mtnt));
    local -= ld_bz * q.mt * q.nt;
    ld_ny = __builtin_amdgcn_readfirstlane(fast_div(local, q.inv_mt));
    ld_mx = local - ld_ny * q.mt;
    ld_kt = 0;
    ld_kt1 = q.kt_total;
    if (q.ksplits > 1) {
      ld_kt = ld_bz * q.per;
      ld_kt1 = ld_kt + q.per < q.kt_total ? ld_kt + q.per : q.kt_total;
    }
    if (tid == 0) {                            // (read by the compute side at least one barrier later)
      int4 d;
      d.x = ld_p;
      d.y = ld_mx | (ld_ny << 16);
      d.z = ld_bz;
      d.w = ld_kt1 - ld_kt;
      *reinterpret_cast<int4*>(items + (ld_seq & 3) * 4) = d;
    }
    ++ld_seq;
    const rmem_linear_args& a = g.p[i];
    const int bzb = q.ksplits > 1 ? 0 : ld_bz;
    int bcol = ld_ny * 128 + wc * 32 + (lane & 31);
    bcol = bcol < q.N ? bcol : q.N - 1;
    // (a problem without a per-column bias: any valid address, the value is not used)
    bsrc = (a.bias && !a.bias_per_row) ? reinterpret_cast<const char*>(a.bias + bzb * a.bsbias + bcol)
                                       : reinterpret_cast<const char*>(q.x[0][0]);
    ld_bz = bzb;                               // (K splits share the operands)
    ld_segment();
  };
  // Requests of one stage: DMA_PER_WAVE per wave (bias, then per plane X / Y rows 8w.. / Y rows 64 + 8w..), always issued
  // (past the last stage: re-reads of the last valid addresses into the dummy KiB) so that every counted wait sees a
  // constant number of requests.
  auto piece = [&](int buf, auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    if constexpr (j == 0) {
      const unsigned db = __builtin_amdgcn_readfirstlane(lds0 + (ld_valid ? Cfg::BIAS + (buf * 8 + wave) * 256 : Cfg::DUMMY));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(db), "v"(bsrc) : "memory");
    } else {
      constexpr int pl = (j - 1) / 3, w = (j - 1) % 3;       // plane; 0 = X, 1 = Y rows 8w.., 2 = Y rows 64 + 8w..
      const int base = ld_valid ? buf * Cfg::STAGE_BYTES + wave * 1024 : Cfg::DUMMY;
      const int on = ld_valid ? 1 : 0;
      constexpr int off = w == 0 ? pl * Cfg::X_BYTES : Cfg::NPL * Cfg::X_BYTES + pl * Cfg::Y_BYTES + (w == 2 ? 8192 : 0);
      const char* gp = w == 0 ? xs[pl] + xo : (w == 1 ? ys[pl] + yo0 : ys[pl] + yo1);
      const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + base + on * off);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(gp) : "memory");
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {      // after a stage's last request: next k-tile, segment or item
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
          } else {                             // (the dummy requests re-read the last stage: never past an operand's end)
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
  frag8_t fa[4][Cfg::NPL], fb[4][Cfg::NPL];   // fragments of the stage whose MFMAs run next
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int pl = 0; pl < Cfg::NPL; ++pl)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fa[ks][pl][e] = 0;
        fb[ks][pl][e] = 0;
      }
  float bc_cur = 0.f, bc_next = 0.f;          // bias of the lane's column: of the stage in the registers / being read

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
    if (compute && cp_left == 0)               // first stage of an item: how many stages it has
      cp_left = __builtin_amdgcn_readfirstlane(items[(cp_seq & 3) * 4 + 3]);
    bc_cur = bc_next;
    const int nbuf = (it + 3) % Cfg::NSTAGE;   // buffer of stage it + 3 = the one stage it was read from
    const int rbuf = (it + 4) % Cfg::NSTAGE;   // buffer of stage it + 1
    const char* st = smem + rbuf * Cfg::STAGE_BYTES;
    static_for<4>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      if (compute) {
        if constexpr (NS == 3) {               // small terms first (gemm_mainloop's order)
          acc = RMEM_MFMA(fa[ks][0], fb[ks][1], acc);
          acc = RMEM_MFMA(fa[ks][1], fb[ks][0], acc);
        }
        acc = RMEM_MFMA(fa[ks][0], fb[ks][0], acc);
      }
      if (refill) {
#pragma unroll
        for (int pl = 0; pl < Cfg::NPL; ++pl) {
          fa[ks][pl] = *reinterpret_cast<const frag8_t*>(st + a_off[ks] + pl * Cfg::X_BYTES);
          fb[ks][pl] = *reinterpret_cast<const frag8_t*>(st + b_off[ks] + pl * Cfg::Y_BYTES);
        }
        if constexpr (ks == 0)
          bc_next = *reinterpret_cast<const float*>(smem + Cfg::BIAS + (rbuf * 8 + wave) * 256 + (lane & 31) * 4);
      }
      // the requests of stage it + 3 (7 = 2 + 2 + 2 + 1, 4 = 1 + 1 + 1 + 1)
      constexpr int per = (Cfg::DMA_PER_WAVE + 3) / 4;
      static_for<per>([&](auto I) {
        constexpr int j = ks * per + decltype(I)::value;
        if constexpr (j < Cfg::DMA_PER_WAVE) piece(nbuf, std::integral_constant<int, j>{});
      });
    });
    advance();
    if (compute && --cp_left == 0) {           // item done: its epilogue, then the next item of this workgroup
      const int4 d = *reinterpret_cast<const int4*>(items + (cp_seq & 3) * 4);
      const int p = __builtin_amdgcn_readfirstlane(d.x), mn = __builtin_amdgcn_readfirstlane(d.y);
      const int bzz = __builtin_amdgcn_readfirstlane(d.z);
      const int m0 = (mn & 0xffff) * 64, n0 = (mn >> 16) * 128;
      const rmem_linear_args& a = g.p[p];
      const int kind = g.q[p].kind;
      const float bc = (a.bias && !a.bias_per_row) ? bc_cur : 0.f;
      const int row0 = m0 + wr * 32 + 4 * (lane >> 5), col = n0 + wc * 32 + (lane & 31);
      const bool full = m0 + 64 <= a.M;        // (uniform)
      const int bz = a.ksplits > 1 ? 0 : bzz;
      if (kind == SK_PARTS) {
        if (full) stream_ep_parts<true>(a, acc, row0, col, bzz, bc);
        else stream_ep_parts<false>(a, acc, row0, col, bzz, bc);
      } else if (kind == SK_F32) {
        if (full) stream_ep_f32<true, false>(a, acc, row0, col, bz, bc);
        else stream_ep_f32<false, false>(a, acc, row0, col, bz, bc);
      } else if (kind == SK_F32_SILU) {
        if (full) stream_ep_f32<true, true>(a, acc, row0, col, bz, bc);
        else stream_ep_f32<false, true>(a, acc, row0, col, bz, bc);
      } else if (kind == SK_BLOCKED) {
        stream_ep_blocked<false>(a, acc, row0, col, bz, bc);
      } else if (kind == SK_BLOCKED_SILU) {
        stream_ep_blocked<true>(a, acc, row0, col, bz, bc);
      } else if (kind == SK_PLANES) {
        if (full) stream_ep_planes<true>(a, acc, row0, col, bz, bc);
        else stream_ep_planes<false>(a, acc, row0, col, bz, bc);
      } else {
        f32x16_t t[1][1];
        t[0][0] = acc;
        linear_epilogue<Cfg>(a, t, m0, n0, bzz, wr, wc, lane);
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the dummy requests of the last stages)
  if (TRACE && tr && threadIdx.x == 0) {
    tr[62] = nst;
    tr[63] = __builtin_readcyclecounter();
  }
}

// ---- host side
static int stream_kind(const rmem_linear_args& a) {
  if (a.bias_per_row) return SK_GENERIC;
  if (a.ksplits > 1) return SK_PARTS;
  if (a.pa_blocked) return a.act == 1 ? SK_BLOCKED_SILU : (a.act == 0 ? SK_BLOCKED : SK_GENERIC);
  if (a.d0 && !a.d1 && !a.pah && !a.pbh && !a.accumulate && a.csplit >= a.N && (a.act == 0 || a.act == 1))
    return a.act == 1 ? SK_F32_SILU : SK_F32;
  if (a.pah && !a.d0 && !a.d1 && a.act == 0) return SK_PLANES;
  return SK_GENERIC;
}

// items of a launch under the streaming kernel's 64 x 128 tiling (args already validated)
static int stream_group(const rmem_linear_args* args, int n, StreamGroup& g) {
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = args[i];
    const rmem_linear_args& a = g.p[i];
    StreamProb& q = g.q[i];
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
    q.kind = stream_kind(a);
    g.tile_start[i] = total;
    total += q.mt * q.nt * (a.ksplits > 1 ? a.ksplits : (a.nbatch > 0 ? a.nbatch : 1));
  }
  g.tile_start[n] = total;
  return total;
}

static int stream_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}

template <int NS, int TRACE>
static int launch_stream(const StreamGroup& g, int total, long long* trace, hipStream_t s) {
  using Cfg = StreamCfg<NS>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_stream_kernel<NS, TRACE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            Cfg::LDS_BYTES);
  const int grid = total < stream_cus() ? total : stream_cus();
  hipLaunchKernelGGL((linear_stream_kernel<NS, TRACE>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, g, trace);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
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
