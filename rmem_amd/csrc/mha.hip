// mha.hip -- 8-head x 32 MultiheadAttention of the AOT block (layers/attention.py:28-81) as a
// fused flash-style kernel: no probability matrix is materialised (with d_head = 32 it would
// be 8x the DeAOT one), the O^T accumulator of a wave is a single 32x32 MFMA tile.
//
// One wave = 32 queries of one head; key tiles of 32.  Both contractions run "swapped":
//   S^T[key][query] = K_tile . Q^T          (A = K rows, B = Q rows, k = head dim 32)
//   O^T[chan][query] += V^T_tile . P^T      (A = V^T rows = channels, B = P^T, k = 32 keys)
// so the softmax statistics of a query live in one lane (+ its lane^32 partner) and the
// accumulator registers of S^T are already the B operand of the second MFMA: accumulator
// register r of lane (query j, half hi) is key (r&3) + 8(r>>2) + 4hi; k-step s, element e of
// the B operand is register 8s+e, i.e. keys {16s+4hi+0..3, 16s+8+4hi+0..3} -- the A operand
// (V^T) is loaded with exactly that key permutation (two 8-byte loads per k-step), so no
// cross-lane shuffle is needed.  The four waves of a block (128 queries of one head) share the
// K / V^T tiles through LDS (64 keys per stage, coalesced 16-byte loads, register prefetch of
// the next stage): fetching the fragments straight from global memory touched 32 cache lines
// per load instruction and made the texture-address path the bound.
// Key splits (flash-decoding style) write unnormalised partials + (max, sum); rmem_mha_combine
// merges them, normalises, averages the per-slot attention mass over heads.
#include "../../include/rmem_hip.h"
#include "rmem_common.h"

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// LDS image of one 64-key stage of head h (shared by the four waves = 128 queries of a block):
//   K   [plane][64 keys][64 B]   rows of 64 B, 16-byte chunk c of row r stored at c ^ ((r >> 2) & 3)
//   V^T [plane][32 chan][136 B]  128 B of keys + 8 B pad: the 8-byte fragment reads of the 32 lanes
//                                of a half-wave (row = lane) then cover all 64 banks exactly once
constexpr int MHA_KPL = 64 * 64, MHA_VROW = 136, MHA_VPL = 32 * MHA_VROW;

// __launch_bounds__(256, 2): two blocks per CU = two waves per SIMD = a budget of 256 registers per lane, which is
// what makes the compiler keep the MFMA accumulators in VGPRs.  With the default (one wave per SIMD, 512 registers)
// it parks S^T and O^T in AGPRs and every 32-key tile pays 32 v_accvgpr_read (16 for the scores, 16 for an O^T that
// is only touched when a maximum moves) on a VALU-bound loop: 180 registers and ~145 VALU per tile against 146
// registers (3 waves per SIMD) and ~95.
template <int NS>
__device__ __forceinline__ void mha_flash_body(const rmem_mha_args& a, const int z) {
  constexpr int NPL = NS == 1 ? 1 : 2;
  constexpr int STAGE_BYTES = NPL * (MHA_KPL + MHA_VPL);
  __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int h = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + j;   // this lane's query (column of S^T / O^T)
  const int sps = a.Npad / 64;                       // 64-key stages per slot
  const int nstages = a.T * sps;
  const int per = (nstages + a.ksplits - 1) / a.ksplits;
  int lo = z * per, hi_s = lo + per;
  if (hi_s > nstages) hi_s = nstages;

  const h16_t* qp[2] = {a.qh, a.ql};
  const h16_t* kp[2] = {a.kh, a.kl};
  const h16_t* vp[2] = {a.vh, a.vl};
  frag8_t qf[NPL][2];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[p][ks] = *reinterpret_cast<const frag8_t*>(qp[p] + (long)q * a.ldq + h * 32 + ks * 16 + hi * 8);

  // staging: thread -> one 16-byte chunk of the K tile and one of the V^T tile, per plane
  const int k_row = tid >> 2, k_ch = tid & 3;        // key 0..63, chunk 0..3 (8 head dims)
  const int v_row = tid >> 3, v_ch = tid & 7;        // channel 0..31, chunk 0..7 (8 keys)
  const int k_dst = k_row * 64 + ((k_ch ^ ((k_row >> 2) & 3)) << 4);
  const int v_dst = v_row * MHA_VROW + v_ch * 16;
  u32x4_t kr[NPL], vr[NPL];
  auto gload = [&](int stage) __attribute__((always_inline)) {
    const int t = stage / sps;
    const int tok0 = (stage - t * sps) * 64;
    const int phys = a.slot_map ? a.slot_map[t] : t;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      kr[p] = *reinterpret_cast<const u32x4_t*>(kp[p] + (long)phys * a.k_slot_stride + (long)(tok0 + k_row) * a.ldk +
                                                h * 32 + k_ch * 8);
      vr[p] = *reinterpret_cast<const u32x4_t*>(vp[p] + (long)phys * a.v_slot_stride + (long)(h * 32 + v_row) * a.ldv +
                                                tok0 + v_ch * 8);
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    char* ks_lds = smem + buf * STAGE_BYTES;
    char* vs_lds = ks_lds + NPL * MHA_KPL;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      *reinterpret_cast<u32x4_t*>(ks_lds + p * MHA_KPL + k_dst) = kr[p];
      u32x2_t w0, w1;
      w0[0] = vr[p][0]; w0[1] = vr[p][1]; w1[0] = vr[p][2]; w1[1] = vr[p][3];
      *reinterpret_cast<u32x2_t*>(vs_lds + p * MHA_VPL + v_dst) = w0;       // rows are 8-byte aligned only
      *reinterpret_cast<u32x2_t*>(vs_lds + p * MHA_VPL + v_dst + 8) = w1;
    }
  };

  f32x16_t o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  const float sl2e = a.scale * 1.44269504088896341f;   // scores are kept in the log2 domain
  float m = -3.0e38f, l = 0.f, lslot = 0.f, bias_t = 0.f;
  int cur_t = -1;
  const bool qvalid = q < a.N;
  float* sml = a.slot_ml ? a.slot_ml + (((long)z * a.Npad + q) * a.heads + h) * a.T * 2 : nullptr;

  if (lo < hi_s) gload(lo);
  for (int stage = lo; stage < hi_s; ++stage) {
    __syncthreads();            // every wave is done with the previous stage
    lstore(0);
    __syncthreads();
    if (stage + 1 < hi_s) gload(stage + 1);          // in flight while this stage is processed
    const int buf = 0;
    const char* ks_lds = smem + buf * STAGE_BYTES;
    const char* vs_lds = ks_lds + NPL * MHA_KPL;
    const int t = stage / sps;
    if (t != cur_t) {
      if (sml && cur_t >= 0 && hi == 0) {
        sml[cur_t * 2] = lslot;
        sml[cur_t * 2 + 1] = m * 0.693147180559945f;     // back to the natural-log domain
      }
      lslot = 0.f;
      cur_t = t;
      bias_t = (a.bias && qvalid) ? a.bias[((long)q * a.heads + h) * a.T + t] * sl2e : 0.f;
    }
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const int tok0 = (stage - t * sps) * 64 + sub * 32;
      // ---- fragments: K rows (keys) for S^T, V^T rows (channels) with the key permutation
      frag8_t kf[NPL][2], vf[NPL][2];
      const int krow = sub * 32 + j;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          kf[p][ks] = *reinterpret_cast<const frag8_t*>(ks_lds + p * MHA_KPL + krow * 64 +
                                                         (((ks * 2 + hi) ^ ((krow >> 2) & 3)) << 4));
          const char* vb = vs_lds + p * MHA_VPL + j * MHA_VROW + sub * 64 + ks * 32 + hi * 8;
          const u32x2_t g0 = *reinterpret_cast<const u32x2_t*>(vb);
          const u32x2_t g1 = *reinterpret_cast<const u32x2_t*>(vb + 16);
          u32x4_t w;
          w[0] = g0[0]; w[1] = g0[1]; w[2] = g1[0]; w[3] = g1[1];
          vf[p][ks] = __builtin_bit_cast(frag8_t, w);
        }
      // ---- S^T = K . Q^T over the head dim (2 k-steps of 16)
      f32x16_t s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (NS == 3) {
          s = RMEM_MFMA(kf[0][ks], qf[1][ks], s);
          s = RMEM_MFMA(kf[1][ks], qf[0][ks], s);
        }
        s = RMEM_MFMA(kf[0][ks], qf[0][ks], s);
      }
      // ---- online softmax for this lane's query, in the log2 domain: y = (s + bias) * scale * log2(e),
      // weights 2^(y - m).  The kernel is VALU-bound, so the token mask is applied only on the
      // tile that contains padding and the accumulator rescale only when some maximum moved.
      // The kernel is VALU-bound (PMC: the VALU is ~85 % busy, 219 instructions per 32-key tile before
      // this form), so the common path is kept to the minimum: the row maximum is taken over the raw
      // scores (scale > 0: max commutes with the affine map), the subtraction of the new maximum
      // rides in the addend of the scaling fma, and the token mask lives on its own branch (written
      // as selects it is if-converted into two v_cndmask per element on EVERY tile).
      const bool padded = tok0 + 32 > a.N;      // wave-uniform
      float tmax = -3.0e38f;
      if (!padded) {
        float tm = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, s[r]);
        tmax = fmaf(tm, sl2e, bias_t);
      } else {
        int tb = tok0 + 4 * hi;
        asm volatile("" : "+v"(tb));            // the 16 token indexes belong to this branch: without the pin they are
#pragma unroll                                  // hoisted in front of it and computed on every tile
        for (int r = 0; r < 16; ++r) {
          const int tok = tb + (r & 3) + 8 * (r >> 2);
          tmax = fmaxf(tmax, tok < a.N ? fmaf(s[r], sl2e, bias_t) : -3.0e38f);
        }
      }
      float tm_lo, tm_hi;
      xor32_pair(tmax, tm_lo, tm_hi);                  // the query's other 16 keys sit in lane ^ 32
      const float m_new = fmaxf(m, fmaxf(tm_lo, tm_hi));
      const float cadd = bias_t - m_new;
      float pv[16];
      f32x2_t ps2 = {0.f, 0.f};
      if (!padded) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2_t pp;
          pp[0] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, cadd));
          pp[1] = __builtin_amdgcn_exp2f(fmaf(s[r + 1], sl2e, cadd));
          pv[r] = pp[0];
          pv[r + 1] = pp[1];
          ps2 += pp;
        }
      } else {
        int tb = tok0 + 4 * hi;
        asm volatile("" : "+v"(tb)::"memory");  // keeps this block a branch, and its token indexes inside it
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tok = tb + (r & 3) + 8 * (r >> 2);
          // (a split may start on an all-padding tile: m_new is still the sentinel there, nothing is valid)
          pv[r] = tok < a.N ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, cadd)) : 0.f;
          ps2[r & 1] += pv[r];
        }
      }
      float ps_lo, ps_hi;
      xor32_pair(ps2[0] + ps2[1], ps_lo, ps_hi);
      const float psum = ps_lo + ps_hi;
      if (__any(m_new != m)) {
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        l *= alpha;
        lslot *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        m = m_new;
      }
      l += psum;
      lslot += psum;
      // ---- P^T as the B operand: k-step s, element e = register 8s+e
      frag8_t pf[NPL][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t wh, wl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // softmax weights <= 1: no saturation needed
          if constexpr (NPL == 2) {
            uint32_t h2, l2;
            split_pair_f16(pv[8 * ks + 2 * e], pv[8 * ks + 2 * e + 1], h2, l2);
            wh[e] = h2;
            wl[e] = l2;
          } else {
            f32x2_t pp;
            pp[0] = pv[8 * ks + 2 * e];
            pp[1] = pv[8 * ks + 2 * e + 1];
            wh[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pp, f16x2_t));
          }
        }
        pf[0][ks] = __builtin_bit_cast(frag8_t, wh);
        if constexpr (NPL == 2) pf[1][ks] = __builtin_bit_cast(frag8_t, wl);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (NS == 3) {
          o = RMEM_MFMA(vf[0][ks], pf[1][ks], o);
          o = RMEM_MFMA(vf[1][ks], pf[0][ks], o);
        }
        o = RMEM_MFMA(vf[0][ks], pf[0][ks], o);
      }
    }
  }
  if (sml && cur_t >= 0 && hi == 0) {
    sml[cur_t * 2] = lslot;
    sml[cur_t * 2 + 1] = m * 0.693147180559945f;
  }
  // ---- partial outputs: O^T rows (channels) of this lane are 4 runs of 4 consecutive channels
  float* op = a.opart + ((long)z * a.Npad + q) * (a.heads * 32) + h * 32 + 4 * hi;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(op + 8 * g) = make_float4(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
  if (hi == 0) {
    float* mlp = a.ml + (((long)z * a.Npad + q) * a.heads + h) * 2;
    mlp[0] = m * 0.693147180559945f;
    mlp[1] = l;
  }
}

template <int NS>
__global__ __launch_bounds__(256, 2) void mha_flash_kernel(rmem_mha_args a) {
  mha_flash_body<NS>(a, blockIdx.z);
}

// Two independent reads of one layer in ONE launch (the AOT block's long-term read and its short-term read both start from
// Q / K of the layer and feed different projections): blockIdx.z < p[0].ksplits serves p[0], the rest p[1].
struct Mha2Args {
  rmem_mha_args p[2];
};
template <int NS>
__global__ __launch_bounds__(256, 2) void mha_flash2_kernel(Mha2Args g) {
  const int za = g.p[0].ksplits;
  const bool first = (int)blockIdx.z < za;            // (ONE inlined body: its LDS image must not be allocated twice)
  const rmem_mha_args& a = first ? g.p[0] : g.p[1];
  mha_flash_body<NS>(a, first ? (int)blockIdx.z : (int)blockIdx.z - za);
}

static int mha_args_ok(const rmem_mha_args& a) {
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) || a.T <= 0 || a.heads <= 0 || a.ksplits <= 0) return 0;
  if (!a.qh || !a.kh || !a.vh || !a.opart || !a.ml) return 0;
  if ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || (a.k_slot_stride % 8) || (a.v_slot_stride % 8)) return 0;
  if (a.nsplit != 1 && a.nsplit != 3) return 0;
  if (a.nsplit == 3 && (!a.ql || !a.kl || !a.vl)) return 0;
  return 1;
}

extern "C" int rmem_mha_flash2(const rmem_mha_args* ap, const rmem_mha_args* bp, void* stream) {
  if (!ap || !bp || !mha_args_ok(*ap) || !mha_args_ok(*bp)) return RMEM_ERR_INVALID;
  if (ap->Npad != bp->Npad || ap->heads != bp->heads || ap->nsplit != bp->nsplit) return RMEM_ERR_INVALID;
  Mha2Args g;
  g.p[0] = *ap;
  g.p[1] = *bp;
  dim3 grid(ap->Npad / 128, ap->heads, ap->ksplits + bp->ksplits);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (ap->nsplit == 3) hipLaunchKernelGGL(mha_flash2_kernel<3>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(mha_flash2_kernel<1>, grid, dim3(256), 0, s, g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_mha_flash(const rmem_mha_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  const rmem_mha_args& a = *ap;
  if (a.N <= 0 || a.Npad < a.N || (a.Npad % 128) || a.T <= 0 || a.heads <= 0 || a.ksplits <= 0) return RMEM_ERR_INVALID;
  if (!a.qh || !a.kh || !a.vh || !a.opart || !a.ml) return RMEM_ERR_INVALID;
  if ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || (a.k_slot_stride % 8) || (a.v_slot_stride % 8)) return RMEM_ERR_INVALID;
  dim3 grid(a.Npad / 128, a.heads, a.ksplits);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.nsplit == 3) {
    if (!a.ql || !a.kl || !a.vl) return RMEM_ERR_INVALID;
    hipLaunchKernelGGL(mha_flash_kernel<3>, grid, dim3(256), 0, s, a);
  } else if (a.nsplit == 1) {
    hipLaunchKernelGGL(mha_flash_kernel<1>, grid, dim3(256), 0, s, a);
  } else {
    return RMEM_ERR_INVALID;
  }
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

// ------------------------------------------------------------------ merge key splits
__device__ __forceinline__ void mha_combine_body(const rmem_mha_combine_args& a, const int q) {
  __shared__ float mh[8], Lh[8];
  const int c = threadIdx.x;           // channel (heads * 32 = 256)
  const int h = c >> 5;
  float m = -3.0e38f;
  for (int s = 0; s < a.ksplits; ++s) m = fmaxf(m, a.ml[(((long)s * a.Npad + q) * a.heads + h) * 2]);
  float L = 0.f, O = 0.f;
  for (int s = 0; s < a.ksplits; ++s) {
    const float* mlp = a.ml + (((long)s * a.Npad + q) * a.heads + h) * 2;
    const float f = expf(mlp[0] - m);
    L += f * mlp[1];
    O += f * a.opart[((long)s * a.Npad + q) * (a.heads * 32) + c];
  }
  const float y = O / L;
  if (a.of32) a.of32[(long)q * a.ldo + c] = y;
  h16_t hi, lo;
  split_f16(y, hi, lo);
  a.oh[(long)q * a.ldo + c] = hi;
  if (a.ol) a.ol[(long)q * a.ldo + c] = lo;
  if (a.mass) {
    if ((c & 31) == 0) {
      mh[h] = m;
      Lh[h] = L;
    }
    __syncthreads();
    if (c < a.T) {
      float acc = 0.f;
      for (int hh = 0; hh < a.heads; ++hh) {
        float sl = 0.f;
        for (int s = 0; s < a.ksplits; ++s) {
          const float* e = a.slot_ml + ((((long)s * a.Npad + q) * a.heads + hh) * a.T + c) * 2;
          if (e[0] != 0.f) sl += e[0] * expf(e[1] - mh[hh]);
        }
        acc += sl / Lh[hh];
      }
      a.mass[(long)q * a.T + c] = acc / (float)a.heads;
    }
  }
}

__global__ __launch_bounds__(256) void mha_combine_kernel(rmem_mha_combine_args a) {
  mha_combine_body(a, blockIdx.x);
}
struct MhaCombine2Args {
  rmem_mha_combine_args p[2];
};
__global__ __launch_bounds__(256) void mha_combine2_kernel(MhaCombine2Args g) {
  const int na = g.p[0].N;
  const bool first = (int)blockIdx.x < na;
  const rmem_mha_combine_args& a = first ? g.p[0] : g.p[1];
  mha_combine_body(a, first ? (int)blockIdx.x : (int)blockIdx.x - na);
}
static int mha_combine_ok(const rmem_mha_combine_args& a) {
  if (a.N <= 0 || a.heads != 8 || a.ksplits <= 0 || !a.opart || !a.ml || !a.oh || a.T > 64) return 0;
  if (a.mass && !a.slot_ml) return 0;
  return 1;
}
extern "C" int rmem_mha_combine2(const rmem_mha_combine_args* ap, const rmem_mha_combine_args* bp, void* stream) {
  if (!ap || !bp || !mha_combine_ok(*ap) || !mha_combine_ok(*bp)) return RMEM_ERR_INVALID;
  MhaCombine2Args g;
  g.p[0] = *ap;
  g.p[1] = *bp;
  hipLaunchKernelGGL(mha_combine2_kernel, dim3(ap->N + bp->N), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_mha_combine(const rmem_mha_combine_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  const rmem_mha_combine_args& a = *ap;
  if (a.N <= 0 || a.heads != 8 || a.ksplits <= 0 || !a.opart || !a.ml || !a.oh || a.T > 64) return RMEM_ERR_INVALID;
  if (a.mass && !a.slot_ml) return RMEM_ERR_INVALID;
  hipLaunchKernelGGL(mha_combine_kernel, dim3(a.N), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}
