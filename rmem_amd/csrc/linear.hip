// linear.hip -- rmem_linear: D = act(X . Y^T + bias) with fused multi-destination
// epilogue (fp32 / split-bf16 planes / residual accumulate).  See include/rmem_hip.h.
#include "../../include/rmem_hip.h"
#include "gemm_core.h"

template <int BM, int BN, int NS>
__global__ __launch_bounds__(256) void linear_kernel(rmem_linear_args a) {
  using Cfg = GemmCfg<BM, BN, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bz = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  RowMajorOperand lx, ly;
  lx.hi = a.xh + bz * a.bsx;
  lx.lo = a.xl ? a.xl + bz * a.bsx : nullptr;
  lx.ld = a.ldx;
  lx.hi2 = a.xh2 ? a.xh2 + bz * a.bsx : nullptr;
  lx.lo2 = a.xl2 ? a.xl2 + bz * a.bsx : nullptr;
  lx.ld2 = a.ldx2;
  lx.kt_split = a.xh2 ? a.kx_split / 64 : (1 << 30);
  lx.row0 = m0;
  lx.rows = a.M;
  ly.hi = a.yh + bz * a.bsy;
  ly.lo = a.yl ? a.yl + bz * a.bsy : nullptr;
  ly.ld = a.ldy;
  ly.hi2 = a.yh2 ? a.yh2 + bz * a.bsy : nullptr;
  ly.lo2 = a.yl2 ? a.yl2 + bz * a.bsy : nullptr;
  ly.ld2 = a.ldy2;
  ly.kt_split = a.yh2 ? a.ky_split / 64 : (1 << 30);
  ly.row0 = n0;
  ly.rows = a.N;

  GemmFrag<Cfg> f;
  f.zero();
  gemm_mainloop<Cfg>(f, lx, ly, 0, a.K / 64, smem);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const float* bias = a.bias ? a.bias + bz * a.bsbias : nullptr;
  float* d0 = a.d0 ? a.d0 + bz * a.bsd : nullptr;
  float* d1 = a.d1 ? a.d1 + bz * a.bsd : nullptr;
  bf16_t* pah = a.pah ? a.pah + bz * a.bspa : nullptr;
  bf16_t* pal = a.pal ? a.pal + bz * a.bspa : nullptr;
  bf16_t* pbh = a.pbh ? a.pbh + bz * a.bspa : nullptr;
  bf16_t* pbl = a.pbl ? a.pbl + bz * a.bspa : nullptr;

#pragma unroll
  for (int tn = 0; tn < Cfg::TN; ++tn) {
    const int col = n0 + frag_col<Cfg>(wc, tn, lane);
    if (col >= a.N) continue;
    const float bcol = (bias && !a.bias_per_row) ? bias[col] : 0.f;
    const float addv = a.addvec ? a.addvec[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + frag_row<Cfg>(wr, tm, r, lane);
        if (row >= a.M) continue;
        float v = f.acc[tm][tn][r] + bcol;
        if (bias && a.bias_per_row) v += bias[row];
        if (a.act == 1) v = silu_f(v);
        if (col < a.csplit) {
          if (d0) {
            float* p = d0 + (long)row * a.ldd0 + col;
            *p = a.accumulate ? (*p + v) : v;
          }
        } else if (d1) {
          float* p = d1 + (long)row * a.ldd1 + (col - a.csplit);
          *p = a.accumulate ? (*p + v) : v;
        }
        if (pah) {
          bf16_t hi, lo;
          split_bf16(v, hi, lo);
          pah[(long)row * a.ldpa + col] = hi;
          if (pal) pal[(long)row * a.ldpa + col] = lo;
        }
        if (pbh) {
          bf16_t hi, lo;
          split_bf16(v + addv, hi, lo);
          pbh[(long)row * a.ldpb + col] = hi;
          if (pbl) pbl[(long)row * a.ldpb + col] = lo;
        }
      }
    }
  }
}

template <int BM, int BN, int NS>
static int launch_linear(const rmem_linear_args& a, hipStream_t s) {
  using Cfg = GemmCfg<BM, BN, NS>;
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.nbatch > 0 ? a.nbatch : 1);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_kernel<BM, BN, NS>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((linear_kernel<BM, BN, NS>), grid, dim3(256), Cfg::LDS_BYTES, s, a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

extern "C" int rmem_linear(const rmem_linear_args* ap, void* stream) {
  if (!ap) return RMEM_ERR_INVALID;
  rmem_linear_args a = *ap;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % 64) != 0) return RMEM_ERR_INVALID;
  if (!a.xh || !a.yh) return RMEM_ERR_INVALID;
  if (a.nsplit != 1 && a.nsplit != 3) return RMEM_ERR_INVALID;
  if (a.nsplit == 3 && (!a.xl || !a.yl)) return RMEM_ERR_INVALID;
  if (a.xh2 && ((a.kx_split % 64) != 0 || a.kx_split <= 0 || a.kx_split >= a.K)) return RMEM_ERR_INVALID;
  if (a.yh2 && ((a.ky_split % 64) != 0 || a.ky_split <= 0 || a.ky_split >= a.K)) return RMEM_ERR_INVALID;
  if ((a.ldx % 8) || (a.ldy % 8)) return RMEM_ERR_INVALID;
  if (a.csplit <= 0 || a.csplit > a.N) a.csplit = a.N;
  int tile = a.tile;
  if (tile == 0) {
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * (a.nbatch > 0 ? a.nbatch : 1);
    tile = blocks128 >= 256 ? 128 : 64;
  }
  if (tile == 128) return a.nsplit == 3 ? launch_linear<128, 128, 3>(a, s) : launch_linear<128, 128, 1>(a, s);
  if (tile == 64) return a.nsplit == 3 ? launch_linear<64, 64, 3>(a, s) : launch_linear<64, 64, 1>(a, s);
  return RMEM_ERR_INVALID;
}
