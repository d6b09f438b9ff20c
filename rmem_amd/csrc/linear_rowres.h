// linear_rowres.h -- LayerNorm + the grouped projections that read the normalised rows, as ONE launch with the row tile
// resident in LDS (included by linear.hip after linear_stream.h, whose specialised epilogues it shares).
//
// Under the streaming kernel (linear_stream.h) the projections at the front of a GPM layer -- Q / relative-position bias /
// temporal-PE bias / V / U (/ ID_U) from norm1(tgt) (and id_norm1(tgt_id)), and QK / V1|V2 / U1|U2 of the gated self
// attention from [norm2(tgt) | id_norm2(tgt_id)] (transformer.py:1104-1123,1223-1232, attention.py:151-172) -- are 432-459
// items of 4-8 stages: every item stages its 64 activation rows again (16 column tiles -> 16 fetches of the same rows),
// every stage moves 48 KB through the L1 / LDS-write path for 96 MFMAs, and the LayerNorm that produced the rows was a launch
// of its own that wrote planes to HBM only for this launch to read them back (profiles/r04m_pmc_x3.json: 64 MB of
// fabric-side traffic per launch for <= 5 MB of operands).
//
// Here a workgroup (8 waves, one per CU) owns a 64-token ROW TILE and a group of eight 32-column units:
//   * prologue: the LayerNorm itself -- each wave normalises 8 rows of every residual stream its units read (x + the
//     split-K partials of the preceding projection, summed in split order: rmem_layernorm_red's arithmetic through the one
//     shared row function ln_row256_vals) and writes the hi / lo planes of the normalised rows straight into LDS
//     ([plane][k-tile of 64][64 rows][128 B], chunks XOR-swizzled as in gemm_core.h); ONE workgroup per row tile and stream
//     (the "owner") also writes the folded residual stream -- into a SECOND buffer, because the other workgroups of the
//     row tile read the unfolded one concurrently -- and, where a later launch needs them, the planes to HBM;
//   * the weights never pass through LDS: they are packed once, at weight-packing time, in MFMA B-fragment order
//     ([32-column unit][16-deep k-step][plane][lane][8 halves]: one contiguous KiB per load instruction, 2 KiB per k-step),
//     and every wave streams the fragments of ITS columns from L2 into a ring of eight k-steps of registers, requested
//     eight k-steps (>= 1500 cycles) ahead -- a weight element is used by exactly one wave of the workgroup, so LDS staging
//     would only add a write and a read;
//   * per k-step a wave reads the A fragments of its rows (2 x 32 rows x 2 planes, ds_read_b128) and issues the same
//     MFMAs in the same order as gemm_mainloop / linear_stream_kernel (hi.lo, lo.hi, hi.hi per 16-deep k-step, k
//     ascending, one accumulator per output tile): results are bit-identical to LayerNorm launch + streaming / tile kernels;
//   * unit = (problem, batch, 32 columns, 64 rows x K <= 256) or (32 rows x K = 512): 96 MFMAs either way.
// Per row tile the activations are fetched once per workgroup of the tile (same XCD: block -> XCD placement keeps a row
// tile's workgroups on one L2), the weights once per row tile.
#pragma once

struct RowresStream {
  const float* x;          // residual stream [N][256] fp32 (mode 0)
  float* xo;               // folded stream out (x + partials), != x; nullptr when nparts == 0
  const float* parts;      // split-K partials of this stream: parts[z * part_stride + row * ldpart + c]
  const float* gamma;
  const float* beta;
  h16_t* oh;               // planes of the normalised rows in HBM: written by the owner (mode 0, may be nullptr) / read (mode 1)
  h16_t* ol;
  long ldo;
};

struct RowresGroup {
  int n, nstreams, mode, N, nparts, J, mt;
  float eps;
  long part_stride, ldpart;
  RowresStream s[2];
  unsigned char need[16];  // per workgroup of a row tile: bit s = normalises stream s
  unsigned char own[16];   //                              bit s = writes stream s back
  unsigned unit[16][8];    // per wave: bits 0-2 problem, 3 valid, 4-5 row sub-tiles (1 / 2), 6 first sub-tile, 7 batch, 8.. column unit
  const h16_t* wpk[8];     // packed weights per problem
  int xk0[8], bxk[8];      // first column of [stream 0 | stream 1] a problem reads, step per batch
  int kind[8];             // stream_kind() of the problem's epilogue
  long long* trace;        // debug (rmem_ln_linear_grouped_trace): shader-clock stamps, trace[(block * 8 + wave) * 8 + k]
  rmem_linear_args p[8];
};

static constexpr int ROWRES_LDS = 2 * 8 * 8192;    // [plane][k-tile 0..7][64 rows][128 B]
static constexpr int ROWRES_RING = 8;

template <int NS, int NSUB>
__device__ __forceinline__ void rowres_unit(const RowresGroup& g, const char* smem, unsigned un, int mt, int lane, const h16_t* wp,
                                            frag8_t (&wf)[ROWRES_RING][NS == 1 ? 1 : 2], float bc, long long* tr) {
  constexpr int NPL = NS == 1 ? 1 : 2;
  const int p = un & 7, sub0 = (un >> 6) & 1, bz = (un >> 7) & 1, u = un >> 8;
  const rmem_linear_args& a = g.p[p];
  const int nks = a.K >> 4;
  const int col = u * 32 + (lane & 31);

  int aoff[NSUB][4];
#pragma unroll
  for (int s = 0; s < NSUB; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) aoff[s][q] = lds_swz((sub0 + s) * 32 + (lane & 31), q * 2 + (lane >> 5));
  f32x16_t acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  const int kt0 = (g.xk0[p] + bz * g.bxk[p]) >> 6;
  const int nit = nks / ROWRES_RING;
#pragma clang loop unroll(disable)
  for (int it = 0; it < nit; ++it) {
    const char* base = smem + (kt0 + it * (ROWRES_RING / 4)) * 8192;
    static_for<ROWRES_RING>([&](auto I) {
      constexpr int i = decltype(I)::value;
      frag8_t af[NSUB][NPL];
#pragma unroll
      for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          af[s][pl] = *reinterpret_cast<const frag8_t*>(base + pl * 65536 + (i >> 2) * 8192 + aoff[s][i & 3]);
      if constexpr (NS == 3) {                 // small terms first (gemm_mainloop's order), the sub-tiles interleaved
#pragma unroll
        for (int s = 0; s < NSUB; ++s) acc[s] = RMEM_MFMA(af[s][0], wf[i][1], acc[s]);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) acc[s] = RMEM_MFMA(af[s][1], wf[i][0], acc[s]);
      }
#pragma unroll
      for (int s = 0; s < NSUB; ++s) acc[s] = RMEM_MFMA(af[s][0], wf[i][0], acc[s]);
      // the fragments eight k-steps on (past the end: the last k-step again, never used)
      int ksn = it * ROWRES_RING + i + ROWRES_RING;
      ksn = ksn < nks ? ksn : nks - 1;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) wf[i][pl] = *reinterpret_cast<const frag8_t*>(wp + (long)ksn * 1024 + pl * 512);
      // (left alone the scheduler sinks all eight refills to the end of the body: the ring then runs one k-step ahead
      // instead of eight)
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  if (tr && lane == 0) tr[4] = __builtin_readcyclecounter();
  const int kind = g.kind[p];
  const bool full = mt * 64 + 64 <= a.M;
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int m0 = mt * 64 + (sub0 + s) * 32;
    const int row0 = m0 + 4 * (lane >> 5);
    if (kind == SK_F32) {
      if (full) stream_ep_f32<true, false>(a, acc[s], row0, col, bz, bc);
      else stream_ep_f32<false, false>(a, acc[s], row0, col, bz, bc);
    } else if (kind == SK_F32_SILU) {
      if (full) stream_ep_f32<true, true>(a, acc[s], row0, col, bz, bc);
      else stream_ep_f32<false, true>(a, acc[s], row0, col, bz, bc);
    } else if (kind == SK_BLOCKED) {
      stream_ep_blocked<false>(a, acc[s], row0, col, bz, bc);
    } else if (kind == SK_BLOCKED_SILU) {
      stream_ep_blocked<true>(a, acc[s], row0, col, bz, bc);
    } else {                                   // SK_PLANES (other destination mixes are refused by the host side)
      if (full) stream_ep_planes<true>(a, acc[s], row0, col, bz, bc);
      else stream_ep_planes<false>(a, acc[s], row0, col, bz, bc);
    }
  }
}

// LayerNorm of the eight rows 8 wave .. 8 wave + 7 of residual stream s (+ its split-K partials, summed in split order) ->
// hi / lo planes of the row tile in LDS; the owner workgroup also writes the folded stream and (optionally) the planes to
// HBM.  Two batches of four rows: every load of a batch is requested before the first use, the four rows' arithmetic is
// branch-free (four independent dependency chains the scheduler interleaves), the global stores follow under ONE uniform
// branch.  Rows beyond N repeat row N - 1 (loads and stores clamped: the same values to the same address).  NP = the number
// of partials when known at compile time (0, 2), -1 = a loop.
template <int NS, int NP>
__device__ __forceinline__ void rowres_ln_stream(const RowresGroup& g, int s, bool wr, int mt, int wave, int lane, char* smem) {
  const RowresStream& st = g.s[s];
  const int N = g.N;
  const float4 gm = *reinterpret_cast<const float4*>(st.gamma + lane * 4);
  const float4 bt = *reinterpret_cast<const float4*>(st.beta + lane * 4);
  const float* px = st.x + lane * 4;
  const float* pp = st.parts + lane * 4;
  const long pstride = g.part_stride, ldpart = g.ldpart;
  const float eps = g.eps;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    long row[4];
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = mt * 64 + wave * 8 + h * 4 + q;
      row[q] = r < N ? r : N - 1;
      v[q] = *reinterpret_cast<const float4*>(px + row[q] * 256);
    }
    if constexpr (NP > 0) {
      float4 w[NP][4];
#pragma unroll
      for (int z = 0; z < NP; ++z)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[z][q] = *reinterpret_cast<const float4*>(pp + z * pstride + row[q] * ldpart);
#pragma unroll
      for (int z = 0; z < NP; ++z)               // split order (rmem_layernorm_red)
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q].x += w[z][q].x; v[q].y += w[z][q].y; v[q].z += w[z][q].z; v[q].w += w[z][q].w; }
    } else if constexpr (NP < 0) {
      for (int z = 0; z < g.nparts; ++z) {
        float4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const float4*>(pp + z * pstride + row[q] * ldpart);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q].x += w[q].x; v[q].y += w[q].y; v[q].z += w[q].z; v[q].w += w[q].w; }
      }
    }
    uint2 vh[4], vl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float y[4];
      ln_row256_gb(v[q], gm, bt, eps, y);
      h16_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_f16(y[e], hi[e], lo[e]);
      vh[q].x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
      vh[q].y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
      vl[q].x = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
      vl[q].y = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
      // k = 256 s + 4 lane: k-tile 4 s + lane / 16, 16-byte chunk (lane % 16) / 2, half (lane & 1)
      char* d = smem + (s * 4 + (lane >> 4)) * 8192 + lds_swz(wave * 8 + h * 4 + q, (lane & 15) >> 1) + (lane & 1) * 8;
      *reinterpret_cast<uint2*>(d) = vh[q];
      if (NS == 3) *reinterpret_cast<uint2*>(d + 65536) = vl[q];
    }
    if (wr) {
      if (NP != 0 && g.nparts > 0 && st.xo) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(st.xo + row[q] * 256 + lane * 4) = v[q];
      }
      if (st.oh) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<uint2*>(st.oh + row[q] * st.ldo + lane * 4) = vh[q];
          if (st.ol) *reinterpret_cast<uint2*>(st.ol + row[q] * st.ldo + lane * 4) = vl[q];
        }
      }
    }
  }
}

template <int NS>
__global__ __launch_bounds__(512) void linear_rowres_kernel(RowresGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (row tile, workgroup of the tile): block b runs on XCD b % 8 (observed, speed only), and the workgroups of a
  // row tile share one XCD so that its rows (and the split-K partials folded into them) reach that L2 once
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int mloc = bi / g.J, j = bi - mloc * g.J;
  const int mt = mloc * 8 + xcd;
  if (mt >= g.mt) return;
  const unsigned un = g.unit[j][wave];
  const int need = g.need[j], own = g.own[j];
  const int N = g.N;
  long long* tr = g.trace ? g.trace + ((long)blockIdx.x * 8 + wave) * 8 : nullptr;
  if (tr && lane == 0) tr[0] = __builtin_readcyclecounter();
  // ---- the first eight k-steps of this wave's weight fragments, requested before anything else: (u, ks, plane) = 1 KiB,
  // lane-linear (a wave without a unit re-reads problem 0's first fragments: the requests are unconditional)
  constexpr int NPL = NS == 1 ? 1 : 2;
  const bool valid = (un >> 3) & 1;
  const h16_t* wp;
  float bc = 0.f;
  {
    const int p = un & 7, bz = (un >> 7) & 1, u = un >> 8;
    const rmem_linear_args& a = g.p[p];
    const int nks = a.K >> 4, nu = (a.N + 31) >> 5;
    wp = g.wpk[p] + ((long)(bz * nu + u) * nks) * 1024 + lane * 8;
    const int col = u * 32 + (lane & 31);
    if (valid && a.bias && !a.bias_per_row) bc = a.bias[bz * a.bsbias + (col < a.N ? col : a.N - 1)];
  }
  frag8_t wf[ROWRES_RING][NPL];
#pragma unroll
  for (int i = 0; i < ROWRES_RING; ++i)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) wf[i][pl] = *reinterpret_cast<const frag8_t*>(wp + (long)i * 1024 + pl * 512);
  if (tr && lane == 0) tr[1] = __builtin_readcyclecounter();

  if (g.mode == 0) {
    // ---- LayerNorm of rows 8 wave .. 8 wave + 7 of every needed stream -> planes in LDS
#pragma unroll 1
    for (int s = 0; s < g.nstreams; ++s) {
      if (!((need >> s) & 1)) continue;
      if (g.nparts == 2) rowres_ln_stream<NS, 2>(g, s, (own >> s) & 1, mt, wave, lane, smem);
      else if (g.nparts == 0) rowres_ln_stream<NS, 0>(g, s, (own >> s) & 1, mt, wave, lane, smem);
      else rowres_ln_stream<NS, -1>(g, s, (own >> s) & 1, mt, wave, lane, smem);
    }
  } else {
    // ---- the planes exist (rmem_layernorm_cn / rmem_layernorm_red wrote them): copy the tile's rows into LDS
#pragma unroll 1
    for (int s = 0; s < g.nstreams; ++s) {
      if (!((need >> s) & 1)) continue;
      const RowresStream& st = g.s[s];
#pragma unroll
      for (int pl = 0; pl < (NS == 1 ? 1 : 2); ++pl) {
        const h16_t* src = pl ? st.ol : st.oh;
        u32x4_t t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {            // chunk id = tid + 512 q: row id >> 5, 16-byte chunk id & 31 of 256 columns
          const int id = tid + 512 * q;
          int row = mt * 64 + (id >> 5);
          row = row < N ? row : N - 1;
          t[q] = *reinterpret_cast<const u32x4_t*>(src + (long)row * st.ldo + (id & 31) * 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int id = tid + 512 * q, r = id >> 5, cg = id & 31;
          *reinterpret_cast<u32x4_t*>(smem + pl * 65536 + (s * 4 + (cg >> 3)) * 8192 + lds_swz(r, cg & 7)) = t[q];
        }
      }
    }
  }
  if (tr && lane == 0) tr[2] = __builtin_readcyclecounter();
  __syncthreads();
  if (tr && lane == 0) tr[3] = __builtin_readcyclecounter();
  if (!valid) return;
  if (((un >> 4) & 3) == 2) rowres_unit<NS, 2>(g, smem, un, mt, lane, wp, wf, bc, tr);
  else rowres_unit<NS, 1>(g, smem, un, mt, lane, wp, wf, bc, tr);
  if (tr && lane == 0) tr[5] = __builtin_readcyclecounter();
}

// ---- host side
static bool rowres_enabled() {
  static const char* e = getenv("RMEM_ROWRES");
  return !(e && e[0] == '0');
}

template <int NS>
static int launch_rowres(const RowresGroup& g, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_rowres_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            ROWRES_LDS);
  const int per_xcd = (g.mt + 7) / 8;          // row tiles of the fullest XCD
  hipLaunchKernelGGL((linear_rowres_kernel<NS>), dim3(8 * per_xcd * g.J), dim3(512), ROWRES_LDS, s, g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

static int ln_linear_grouped_impl(const rmem_rowres_stream* streams, int32_t nstreams, int32_t mode, int32_t N, int32_t nparts,
                                  int64_t part_stride, int64_t ldpart, float eps, const rmem_rowres_problem* probs, int32_t n,
                                  long long* trace, void* stream) {
  if (!streams || !probs || nstreams < 1 || nstreams > 2 || n < 1 || n > 8 || N <= 0 || nparts < 0 || (mode != 0 && mode != 1))
    return RMEM_ERR_INVALID;
  if (rmem::current_recorder() || !rowres_enabled()) return RMEM_ERR_INVALID;
  RowresGroup gl;
  memset(&gl, 0, sizeof(gl));
  gl.n = n; gl.nstreams = nstreams; gl.mode = mode; gl.N = N; gl.nparts = nparts; gl.eps = eps;
  gl.part_stride = part_stride; gl.ldpart = ldpart;
  gl.mt = (N + 63) / 64;
  gl.trace = trace;
  for (int s = 0; s < nstreams; ++s) {
    const rmem_rowres_stream& in = streams[s];
    RowresStream& st = gl.s[s];
    st.x = in.x; st.xo = in.xo; st.parts = in.parts; st.gamma = in.gamma; st.beta = in.beta;
    st.oh = in.oh; st.ol = in.ol; st.ldo = in.ldo;
    if (mode == 0) {
      if (!st.x || !st.gamma || !st.beta) return RMEM_ERR_INVALID;
      if (nparts > 0 && (!st.parts || !st.xo || st.xo == st.x || (ldpart % 4) || (part_stride % 4))) return RMEM_ERR_INVALID;
      if (st.oh && (st.ldo % 4)) return RMEM_ERR_INVALID;
    } else if (!st.oh || (st.ldo % 8)) {
      return RMEM_ERR_INVALID;
    }
  }
  const int ns = probs[0].lin.nsplit;
  if (ns != 1 && ns != 3) return RMEM_ERR_INVALID;
  if (ns == 3 && mode == 1 && (!gl.s[0].ol || (nstreams > 1 && !gl.s[1].ol))) return RMEM_ERR_INVALID;
  // units, ordered by the streams they read so that most workgroups normalise one stream only
  struct U { unsigned un; int mask; };
  U units[16 * 8];
  int nun = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int want = pass == 0 ? 1 : (pass == 1 ? 3 : 2);
    for (int i = 0; i < n; ++i) {
      const rmem_rowres_problem& pr = probs[i];
      const rmem_linear_args& a = pr.lin;
      if (pass == 0) {
        gl.p[i] = a;
        gl.wpk[i] = pr.wpk;
        gl.xk0[i] = pr.xk0;
        gl.bxk[i] = pr.bxk;
        const int nb = a.nbatch > 1 ? a.nbatch : 1;
        if (!pr.wpk || a.M != N || a.N <= 0 || a.K <= 0 || (a.K % 128) || a.K > 512 || a.nsplit != ns || nb > 2 || a.ksplits > 1 ||
            (pr.xk0 % 64) || (pr.bxk % 64) || pr.xk0 < 0 || pr.xk0 + (nb - 1) * pr.bxk + a.K > 256 * nstreams || a.N > (1 << 20))
          return RMEM_ERR_INVALID;
        if (gl.p[i].csplit <= 0 || gl.p[i].csplit > a.N) gl.p[i].csplit = a.N;
        if (a.pa_blocked && (!a.pah || a.d0 || a.d1 || a.pbh)) return RMEM_ERR_INVALID;
        gl.kind[i] = stream_kind(gl.p[i]);
        if (gl.kind[i] == SK_GENERIC || gl.kind[i] == SK_PARTS) return RMEM_ERR_INVALID;
      }
      const int nb = a.nbatch > 1 ? a.nbatch : 1;
      for (int b = 0; b < nb; ++b) {
        const int k0 = pr.xk0 + b * pr.bxk, k1 = k0 + a.K;
        const int mask = (k0 < 256 ? 1 : 0) | (k1 > 256 ? 2 : 0);
        if (mask != want) continue;
        const int nu = (a.N + 31) / 32;
        const int nsub = a.K <= 256 ? 2 : 1;
        for (int u = 0; u < nu; ++u)
          for (int sub0 = 0; sub0 < 2; sub0 += nsub) {
            if (nun >= 16 * 8) return RMEM_ERR_INVALID;
            units[nun].un = (unsigned)i | 8u | ((unsigned)nsub << 4) | ((unsigned)sub0 << 6) | ((unsigned)b << 7) | ((unsigned)u << 8);
            units[nun].mask = mask;
            ++nun;
          }
      }
    }
  }
  gl.J = (nun + 7) / 8;
  for (int k = 0; k < nun; ++k) {
    gl.unit[k / 8][k % 8] = units[k].un;
    gl.need[k / 8] |= (unsigned char)units[k].mask;
  }
  for (int s = 0; s < nstreams; ++s) {           // owner of a stream: the first workgroup that normalises it
    int o = -1;
    for (int j = 0; j < gl.J && o < 0; ++j)
      if ((gl.need[j] >> s) & 1) o = j;
    if (o < 0) { o = 0; gl.need[0] |= (unsigned char)(1 << s); }
    gl.own[o] |= (unsigned char)(1 << s);
  }
  hipStream_t hs = static_cast<hipStream_t>(stream);
  return ns == 3 ? launch_rowres<3>(gl, hs) : launch_rowres<1>(gl, hs);
}

extern "C" int rmem_ln_linear_grouped(const rmem_rowres_stream* streams, int32_t nstreams, int32_t mode, int32_t N, int32_t nparts,
                                      int64_t part_stride, int64_t ldpart, float eps, const rmem_rowres_problem* probs, int32_t n,
                                      void* stream) {
  return ln_linear_grouped_impl(streams, nstreams, mode, N, nparts, part_stride, ldpart, eps, probs, n, nullptr, stream);
}

// Debug aid (tools/kbench_rowres.py): the same launch with shader-clock stamps per wave, trace[(block * 8 + wave) * 8 + k]:
// [0] start, [1] first weight fragments requested, [2] LayerNorm / plane copy done, [3] barrier passed, [4] MFMAs issued,
// [5] end.  trace must hold 64 int64 per workgroup of the launch (8 * ceil(row tiles / 8) * workgroups per row tile <= 8 *
// ceil(N / 512) * 16), zeroed by the caller.
extern "C" int rmem_ln_linear_grouped_trace(const rmem_rowres_stream* streams, int32_t nstreams, int32_t mode, int32_t N,
                                            int32_t nparts, int64_t part_stride, int64_t ldpart, float eps,
                                            const rmem_rowres_problem* probs, int32_t n, int64_t* trace, void* stream) {
  if (!trace) return RMEM_ERR_INVALID;
  return ln_linear_grouped_impl(streams, nstreams, mode, N, nparts, part_stride, ldpart, eps, probs, n,
                                reinterpret_cast<long long*>(trace), stream);
}
