/* rmem_hip.h -- C ABI of librmem_hip.so: the MI355X (gfx950) kernels of the RMem /
 * DeAOT hot path.
 *
 * The reference (Restricted-Memory/RMem) is pure PyTorch and has no FFI for this path
 * (SURVEY.md section 8b); these entry points are what a binding of the reference's
 * attention layers would call instead of their PyTorch bodies.  Every declaration
 * cites the reference code it replaces (paths relative to /root/reference/aot_plus/).
 *
 * Conventions
 *   - every pointer is a device (HBM) pointer unless marked "host";
 *   - no hidden allocation, no synchronisation: the caller owns all buffers and
 *     workspaces; kernels are enqueued on `stream` (a hipStream_t passed as void*), so
 *     calls are hipGraph-capturable.  The library reads no environment variable; its
 *     only process-wide state is the set of tuning switches of rmem_configure() below
 *     (and the per-thread launch recorder, rmem_rec_*);
 *   - return value: 0 = enqueued, <0 = error (RMEM_ERR_*); no exceptions cross the ABI;
 *   - "planes": an fp32 tensor carried as two fp16 tensors hi = fp16(x),
 *     lo = fp16(x - hi), values beyond +-65504 saturated (rmem_f16 = raw IEEE fp16 bits).
 *     `nsplit` = 3 multiplies hi*lo + lo*hi + hi*hi on the 16-bit MFMA pipe
 *     (fp32-class accuracy), `nsplit` = 1 multiplies hi*hi only (plain fp16) and ignores
 *     the lo pointers;
 *   - tokens are row-major over the feature map, p = y*w + x (layers/basic.py:73-77).
 */
#ifndef RMEM_HIP_H
#define RMEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t rmem_f16;

#define RMEM_OK 0
#define RMEM_ERR_INVALID (-1)
#define RMEM_ERR_LAUNCH (-2)

/* ABI version (bumped on any signature change). */
int rmem_abi_version(void);

/* Process-wide tuning / debug switches (every one selects between kernels that compute the same results bit for bit, or
 * serves a debug entry point).  Returns RMEM_ERR_INVALID for an unknown name or a value outside the range; takes effect for
 * launches issued afterwards.  Names (range, default):
 *   "linear_tiles"  (0-1, 0)   1: tile-per-workgroup projection kernels for every rmem_linear[_grouped] launch
 *   "stream_var"    (1-4, 1)   rmem_linear_trace only: 2 no operand requests, 3 no MFMAs, 4 no fragment reads (timings, wrong results)
 *   "dw_rows"       (0-4, 2)   rmem_dwconv5x5_split[2]: output rows per thread (0 = the one-row kernel)
 *   "dw_rx" (6-12, 9), "dw_v" (1-4, 1)   the one-row kernel's tokens / channels per thread; "dw_grid_order" (0-1, 0) plain block order
 *   "ida_tokens" (1-2, 1), "ida_unroll" (8-32, 16)   rmem_id_assign
 *   "read_var"      (0-31, 0)  rmem_attn_read_trace only: experiment mask of the tracing kernel */
int rmem_configure(const char *name, int64_t value);

/* How host threads of this process wait for `device` (hipSetDeviceFlags): blocking != 0 = sleep on the interrupt
 * (hipDeviceScheduleBlockingSync), 0 = the runtime's default, which spins on a core.  The engine's host thread runs up
 * to `long_term_mem_gap` frames ahead of the GPU and waits for it at every long-term memory update
 * (the reference waits in every frame: `.item()` / `.cpu()` in engines/aot_engine.py:350-356 and
 * networks/layers/transformer.py:880-991) -- with the default that wait costs one core per rank, all the time.
 * Call before the process first touches the device for the full effect (later calls reach only streams created
 * afterwards).  Leaves the calling thread's current device unchanged.  OPT-IN (the Python host calls it only with
 * RMEM_BLOCKING_WAIT=1): with two processes sharing one GPU the first MIOpen convolution never returned under this flag
 * on ROCm 7.2; the host side's one long wait polls with a 0.2 ms sleep instead (rmem_amd/hip.py: wait_event). */
int rmem_set_host_wait(int32_t device, int32_t blocking);

/* ------------------------------------------------------------------ linear layers
 * D = act(X . Y^T + bias) on the fp16 MFMA pipe (v_mfma_f32_32x32x16_f16).  X is [M][K], Y is [N][K] (both
 * K-contiguous planes), D is [M][N].  Either operand may be continued along K by a
 * second source (x2/y2) after k_split elements (concatenated inputs).  With
 * bias_per_row the roles are swapped (X = weight, Y = activation) and D is the
 * transposed output, which is how channel-major memories (V^T) are produced.
 * Replaces the nn.Linear / 1x1-conv calls of GatedPropagationModule.forward
 * (layers/transformer.py:1106-1123,1238-1244), GatedPropagation.forward
 * (layers/attention.py:151-172,209) and relative_emb_k (layers/attention.py:314).
 */
typedef struct {
  const rmem_f16 *xh, *xl; int64_t ldx;   /* X planes, first K segment              */
  const rmem_f16 *xh2, *xl2; int64_t ldx2; /* second K segment (or NULL)             */
  int32_t kx_split;                        /* elements of K served by the first X source */
  const rmem_f16 *yh, *yl; int64_t ldy;
  const rmem_f16 *yh2, *yl2; int64_t ldy2;
  int32_t ky_split;
  int32_t M, N, K;                         /* K, k*_split multiples of 64            */
  const float *bias; int32_t bias_per_row; /* bias[N] (or bias[M] when per row), may be NULL */
  int32_t act;                             /* 0 none, 1 SiLU (x*sigmoid(x), attention.py:89) */
  float *d0; int64_t ldd0;                 /* fp32 out for columns [0, csplit)       */
  float *d1; int64_t ldd1;                 /* fp32 out for columns [csplit, N)       */
  int32_t csplit;                          /* = N when there is one destination       */
  int32_t accumulate;                      /* D += result (fused residual add)       */
  rmem_f16 *pah, *pal; int64_t ldpa;      /* planes of the result (may be NULL)      */
  rmem_f16 *pbh, *pbl; int64_t ldpb;      /* planes of result + addvec[col] (may be NULL) */
  const float *addvec;
  int32_t nbatch;                          /* gridDim.z; strides below in elements    */
  int64_t bsx, bsy, bsd, bsbias, bspa;
  int32_t nsplit;                          /* 1 or 3                                   */
  int32_t tile;                            /* 0 auto / 256: the streaming kernel (one persistent workgroup per CU, 64 x 128 tiles,
                                              operands by LDS-DMA three stages deep); 64 (64x64; 128 is accepted as 64), 192 (64 rows x
                                              128 cols): the tile-per-workgroup kernels.  Results are bit-identical. */
  int32_t ksplits;                         /* >1: split K; raw partial sums (+bias in split 0) go to
                                              parts[split][M][N] instead of the outputs above; the
                                              consumer sums them in split order (rmem_layernorm_red) */
  float *parts; int64_t part_stride;       /* elements between splits (>= M*N)          */
  int32_t pa_blocked;                      /* != 0: pah/pal are written "blocked-16" over rows: element (row, col)
                                              at ((row/16)*ldpa + col)*16 + row%16 (ldpa = columns of the blocked
                                              tensor; a column window is selected by offsetting pah/pal by 16*col0
                                              elements); the V operand layout of rmem_attn_read; only output allowed */
  int32_t d0_cs;                           /* column stride of d0 in elements (0 or 1 = contiguous).  With ldd0 = 225 and
                                              d0_cs = 226 element (q, o) lands at 225*(q+o) + o: the relative-bias matrix
                                              stored by anti-diagonals, which is what makes its gather in the windowed
                                              read coalesced (rmem_read_args.rcs) */
} rmem_linear_args;

int rmem_linear(const rmem_linear_args *a, void *stream);

/* Up to 8 independent problems (same nsplit, 64x64 tiles) in ONE launch: the small projections
 * of one LSTT stage share their input and individually cannot fill 256 CUs. */
int rmem_linear_grouped(const rmem_linear_args *args, int32_t n, void *stream);

/* Debug aid (tools/kbench_gemm.py): the streaming kernel for `n` problems with shader-clock stamps of every workgroup's
 * wave 0 written to trace[workgroup][64] ([0] start, [1] first requests out, [2 + 2 s] / [3 + 2 s] stage s landed / issued,
 * [62] stages, [63] end).  trace must hold 64 int64 per CU.  Split precision (nsplit = 3) only. */
int rmem_linear_trace(const rmem_linear_args *args, int32_t n, int64_t *trace, void *stream);

/* ------------------------------------------------------------------ fused memory read
 * The memory read of GatedPropagation.forward / LocalGatedPropagation.forward (layers/attention.py:
 * 174-209 and 289-358, call sites layers/transformer.py:1183, 1199, 1227) as ONE flash-style
 * launch: S = scale*(Q.K^T + bias), softmax, O = P.V per key split; the probability matrix
 * stays in LDS.  rmem_attn_read_combine merges the splits, normalises, gates with U and emits the
 * per-slot attention mass (record_attn_weight, transformer.py:1186-1192).
 *
 * mode 0 ("bank", long-term / self): keys are T logical slots of the ring bank,
 *   slot t lives at physical slot slot_map[t]; the temporal positional embedding
 *   (layers/transformer.py:1140-1172) enters as bias[q][t] = (Q[q]+cur_pe).mem_pe[row(t)]
 *   so the stored keys stay PE-free.
 * mode 1 ("window", short-term): one slot (the previous frame); key k is visible to
 *   query q iff |ky-qy| <= 7 and |kx-qx| <= 7 and it gets the relative bias
 *   R[q][(ky-qy+7)*15 + (kx-qx+7)] (layers/attention.py:305-346: out-of-image keys are
 *   the -1e8 entries, which softmax turns into exact zeros).
 *
 * Layouts: K planes [slot][Npad][128]; Q planes [Npad][128]; Npad = N rounded up to 128.  V is "blocked-16":
 * planes [slot][Npad/16][ncols][16] (element (key k, column c) at ((k/16)*ncols + c)*16 + k%16),
 * written by rmem_linear with pa_blocked = 1, so that an MFMA B fragment of 32 columns is one
 * contiguous KiB.  ncols must be 1024 (the eight waves of a unit own 128 columns of [V | ID_V] each).
 * Split precision (hi/lo planes, 3 products) only.

 */
typedef struct {
  int32_t mode;                            /* 0 bank, 1 window                         */
  const rmem_f16 *qh, *ql;                 /* Q planes [Npad][128]                      */
  const rmem_f16 *kh, *kl; int64_t k_slot_stride;   /* K planes [slot][Npad][128]       */
  const rmem_f16 *vh, *vl; int64_t v_slot_stride;   /* V planes, blocked-16             */
  const int32_t *slot_map;                 /* device [T] logical -> physical slot, or NULL */
  int32_t T, N, Npad, ncols;
  float scale;
  const float *bias;                       /* mode 0: [N][T] or NULL                    */
  const float *R; int32_t ldr; int32_t h, w;  /* mode 1: relative bias, element (q, o) at R[q*ldr + o*rcs] */
  int32_t rcs;                             /* column stride of R (0 = 1).  ldr = 225, rcs = 226 = stored by anti-diagonals:
                                              the 32 queries of a wave (consecutive x) then read 32 CONSECUTIVE floats for a
                                              given key instead of 32 different cache lines */
  int32_t ksplits;                         /* <= 32                                     */
  float *part;                             /* [ksplits][Npad][ncols] un-normalised partial O */
  float *ml;                               /* [ksplits][Npad][2]  (running max, row sum) */
  float *lslot;                            /* [ksplits][Npad][T][2] per-slot (sum, max at that time) or NULL */
  int32_t *sched;                          /* NULL, or 2 device ints the caller zeroed ONCE (rmem_attn_read2, field of `a`): when the
                                              launch holds more units than the device has CUs, one workgroup per CU takes its
                                              first unit by index and every further one from this counter, longest first,
                                              instead of leaving the surplus to the hardware's per-XCD dispatch order; the
                                              last workgroup out zeroes the two ints again, and the library zeroes them on the launch stream in front of every such
                                              launch.  One launch at a time per buffer. */
  int32_t nfull, pf;                       /* mode 0, rmem_attn_read2 / rmem_attn_read: nfull > 0 = UNEVEN key splits: the first nfull splits
                                              hold pf 64-key tiles each, the other ksplits - nfull share the rest evenly (0 = even).
                                              rmem_attn_read2 launches the short pieces last, behind the windowed units, so that
                                              they run on the CUs the windowed units leave early.  Direct launches only. */
  const float *gate; int64_t ldgate;       /* with ksplits == 1 and gout != NULL the read needs no combine: its units write the normalised, */
  float *gout; int64_t ldgout;             /* gated aggregate gout[q][c] = gate[q][c] * O[q][c] / l[q] themselves (attention.py:206-209) -- what
                                              rmem_attn_read_combine computes from one split, operation for operation (bit-identical) --
                                              and no partial O (part is not written; ml still is).  The windowed read of a layer in one
                                              split is as long as a long-term unit (the 15 x 15 band of a 64-query tile spans ~15 key tiles):
                                              no partial round trip for it.  Ignored when ksplits > 1; lslot must be NULL (no mass). */
  float *dbg_logits; int64_t dbg_ld;       /* debug, rmem_attn_read_trace only (ignored by every other entry): when non-NULL every
                                              pre-softmax logit scale*(Q.K + bias) (mode 0) / scale*Q.K + R (mode 1, keys inside the
                                              window) is written to dbg_logits[q*dbg_ld + t*N + key] -- the tensor the reference
                                              takes its softmax of (layers/attention.py:184, :344); other elements are left alone */
} rmem_read_args;

int rmem_attn_read(const rmem_read_args *a, void *stream);
/* Debug aid (tools/kbench_read.py): the same launch with shader-clock stamps per workgroup in trace[block][64]
 * ([0] start, [1] Q and the first K tiles staged, [2] tile loop done, [3] end, [4+w] / [12+w] / [20+w] cycles of wave w in its
 * score / P.V phases / at the interval barriers, [28] key tiles of the unit, [32+w] HW_REG_HW_ID of wave w,
 * [40+w] cycles at the top of the iterations, [48] / [49] start / end on the device-wide 100 MHz counter); trace holds 8 * ceil(units / 8) * 64 int64. */
int rmem_attn_read_trace(const rmem_read_args *a, int64_t *trace, void *stream);
/* the bank read (a: mode 0) and the windowed read (b: mode 1) of one GPM layer in one launch */
int rmem_attn_read2(const rmem_read_args *a, const rmem_read_args *b, void *stream);

typedef struct {
  int32_t T, N, Npad, ncols, ksplits;
  const float *part; const float *ml; const float *lslot;
  const float *U; int64_t ldu;             /* gate [N][ncols] (attention.py:206)        */
  float *G; int64_t ldg;                   /* out: gated aggregate [N][ncols] fp32      */
  float *mass;                             /* out (may be NULL): [N][T]                 */
} rmem_read_combine_args;

int rmem_attn_read_combine(const rmem_read_combine_args *a, void *stream);
/* two reads of one layer in one launch (long-term bank + windowed short-term) */
int rmem_attn_read_combine2(const rmem_read_combine_args *a, const rmem_read_combine_args *b, void *stream);

/* ------------------------------------------------------------------ AOT multi-head attention
 * MultiheadAttention.forward of the AOT block (layers/attention.py:28-81; 8 heads x 32,
 * call sites layers/transformer.py:561, 632-635, 656-662) as a fused flash-style kernel with
 * key splits, followed by rmem_mha_combine (merge splits, normalise, head-averaged per-slot
 * attention mass = record_attn_weight, transformer.py:636-644).
 * Layouts: Q planes [Npad][ldq] (head h = columns 32h..32h+31), K planes [slot][Npad][ldk],
 * V^T planes [slot][heads*32][ldv] (ldv >= Npad, keys contiguous).  bias: [N][heads][T]
 * temporal-PE bias or NULL.  opart: [ksplits][Npad][heads*32] fp32, ml: [ksplits][Npad][heads][2],
 * slot_ml: [ksplits][Npad][heads][T][2] zero-filled by the caller (or NULL when no mass is needed).
 */
typedef struct {
  const rmem_f16 *qh, *ql; int64_t ldq;
  const rmem_f16 *kh, *kl; int64_t k_slot_stride, ldk;
  const rmem_f16 *vh, *vl; int64_t v_slot_stride, ldv;
  const int32_t *slot_map; int32_t T, N, Npad, heads;
  float scale;                              /* 1/sqrt(32)                               */
  const float *bias;
  int32_t ksplits;
  float *opart; float *ml; float *slot_ml;
  int32_t nsplit;
} rmem_mha_args;

int rmem_mha_flash(const rmem_mha_args *a, void *stream);
/* two independent reads in one launch (the AOT block's long-term and short-term reads of a layer, transformer.py:632-635 and
 * :656-662: same Npad, heads, nsplit; workspaces of their own) */
int rmem_mha_flash2(const rmem_mha_args *a, const rmem_mha_args *b, void *stream);

typedef struct {
  int32_t N, Npad, heads, T, ksplits;
  const float *opart; const float *ml; const float *slot_ml;
  rmem_f16 *oh, *ol; float *of32; int64_t ldo;   /* out [N][heads*32]: planes (+ optional fp32) */
  float *mass;                                     /* [N][T] or NULL                            */
} rmem_mha_combine_args;

int rmem_mha_combine(const rmem_mha_combine_args *a, void *stream);
int rmem_mha_combine2(const rmem_mha_combine_args *a, const rmem_mha_combine_args *b, void *stream);

/* ------------------------------------------------------------------ pointwise / norms */
/* LayerNorm over C=256 channels -> planes (+ optional fp32); nn.LayerNorm of
 * layers/transformer.py:1104,1120,1223-1224 and models/deaot.py:41. */
int rmem_layernorm_split(const float *x, int64_t ldx, const float *gamma, const float *beta,
                         int32_t N, int32_t C, float eps, rmem_f16 *oh, rmem_f16 *ol,
                         int64_t ldo, float *of32, int64_t ldof, void *stream);

/* LayerNorm with fused adds: y = LN(x + x2) * gamma + beta + post  (x2 / post may be NULL).
 * AOT: q = k = norm1(tgt) + pos (transformer.py:558-560), norm4(local_K + curr_K) (:656-662),
 * norm2(tgt) + id_emb as the input of linear_V (:586, :277-280). */
int rmem_layernorm_ex(const float *x, int64_t ldx, const float *x2, int64_t ldx2, const float *gamma,
                      const float *beta, int32_t N, int32_t C, float eps, const float *post,
                      int64_t ldpost, rmem_f16 *oh, rmem_f16 *ol, int64_t ldo, float *of32,
                      int64_t ldof, void *stream);

/* Up to four rmem_layernorm_ex problems over the same N rows in ONE launch (per problem the same arithmetic, bit for bit).
 * The AOT block normalises one input twice -- q = k = norm1(tgt) + pos beside v = norm1(tgt), transformer.py:558-561 -- and
 * two inputs with one norm -- norm4(local_K + curr_K), norm4(local_V + curr_V), :656-662. */
typedef struct {
  const float *x; int64_t ldx;
  const float *x2; int64_t ldx2;           /* optional second summand                  */
  float *sum_out; int64_t ldsum;           /* optional: x + x2 written here (may be x itself: the residual add tgt += tgt3 in
                                              front of norm3, transformer.py:680-683, without a launch of its own) */
  const float *gamma, *beta;
  const float *post; int64_t ldpost;       /* optional: added after the affine          */
  rmem_f16 *oh, *ol; int64_t ldo;          /* planes out (may be NULL)                  */
  float *of32; int64_t ldof;               /* fp32 out (may be NULL)                    */
} rmem_ln_args;
int rmem_layernorm_multi(const rmem_ln_args *p, int32_t n, int32_t N, int32_t C, float eps, void *stream);

/* Residual reduce + LayerNorm: x[row][:] += sum_z parts[z][row][:]  (z in split order, written
 * back to x), then y = LN(x)*gamma+beta -> planes (+ optional fp32).  Consumes the split-K
 * partials of the projection GEMM that precedes norm2/id_norm2/norm1/id_norm1
 * (layers/transformer.py:1212-1224, 1231-1232).  nparts = 0 is a plain LayerNorm. */
int rmem_layernorm_red(float *x, int64_t ldx, const float *parts, int32_t nparts, int64_t part_stride,
                       int64_t ldpart, const float *gamma, const float *beta, int32_t N, int32_t C,
                       float eps, rmem_f16 *oh, rmem_f16 *ol, int64_t ldo, float *of32, int64_t ldof,
                       void *stream);

/* LayerNorm of the first layer straight from the encoder feature map (replaces bchw_2_lbc + norm1 of layer 0,
 * utils/tensor.py:3-6 + layers/transformer.py:1104): src_cn is channel-major [C = 256][ld_src >= N] fp32; writes the
 * token-major residual stream x [N][256], zeroes `zero` [N][256] when given (the ID stream starts a frame at 0,
 * transformer.py:779) and the normalised planes oh / ol [N][ldo].  Per row the arithmetic of rmem_layernorm_red. */
int rmem_layernorm_cn(const float *src_cn, int64_t ld_src, float *x, float *zero, const float *gamma, const float *beta,
                      int32_t N, int32_t C, float eps, rmem_f16 *oh, rmem_f16 *ol, int64_t ldo, void *stream);

/* Two such problems of the same shape in one launch: norm1 / id_norm1 and norm2 / id_norm2 of a
 * GPM layer (layers/transformer.py:1104, 1120, 1223-1224), each with its own residual stream,
 * partials, affine parameters and output planes. */
int rmem_layernorm_red2(float *x0, float *x1, int64_t ldx, const float *parts0, const float *parts1,
                        int32_t nparts, int64_t part_stride, int64_t ldpart, const float *gamma0,
                        const float *beta0, const float *gamma1, const float *beta1, int32_t N, int32_t C,
                        float eps, rmem_f16 *oh0, rmem_f16 *ol0, int64_t ldo0, rmem_f16 *oh1,
                        rmem_f16 *ol1, int64_t ldo1, void *stream);

/* planes [N][C] (ld) -> transposed planes [C][ldo]  (AOT: V operand of the short-term attention) */
int rmem_transpose_planes(const rmem_f16 *ih, const rmem_f16 *il, int64_t ld, int32_t N, int32_t C,
                          rmem_f16 *oh, rmem_f16 *ol, int64_t ldo, void *stream);

/* dst (fp32, optional, may alias a) = a + b ; planes of the sum (optional).  n elements. */
int rmem_add_split(const float *a, const float *b, int64_t n, float *dst, rmem_f16 *oh,
                   rmem_f16 *ol, void *stream);

/* Up to eight such sums of `nelem` elements each in one launch (the AOT block's memory update: curr_V + id_emb and
 * local_V + id_emb of every layer, transformer.py:277-287). */
typedef struct {
  const float *a, *b;                      /* b may be NULL                             */
  float *dst;                              /* fp32 sum (optional, may alias a)          */
  rmem_f16 *oh, *ol;                       /* planes of the sum (optional)              */
} rmem_add_args;
int rmem_add_split_multi(const rmem_add_args *p, int32_t n, int64_t nelem, void *stream);

/* GNActDWConv2d front half (layers/basic.py:27-32): GroupNorm(groups) over token-major
 * [N][C] (statistics over C/groups channels x N tokens) followed by exact GELU -> fp32.
 * ws: >= 2*16*groups doubles. */
int rmem_gn_gelu_tokens(const float *x, int32_t N, int32_t C, int32_t groups, const float *gamma,
                        const float *beta, float eps, double *ws, float *y, void *stream);

/* temporal-PE bias per head: bias[q][h][t] = sum_{c in head h} (Q[q][c]+cur_pe[c]) * mem_pe[pe_row[t]][c] */
int rmem_pe_bias_heads(const float *Q, int64_t ldq, const float *cur_pe, const float *mem_pe,
                       const int32_t *pe_row_host, int32_t T, int32_t N, int32_t heads, float *bias,
                       void *stream);

/* Depth-wise 5x5, pad 2, no bias, on token-major [h*w][C] (layers/basic.py:38-57);
 * wt is [25][C] (tap-major).  Output planes. */
int rmem_dwconv5x5_split(const float *g, int64_t ldg, const float *wt, int32_t h, int32_t w,
                         int32_t C, rmem_f16 *oh, rmem_f16 *ol, int64_t ldo, void *stream);

/* Two maps of the same geometry in one launch (the gated long-term and short-term aggregates). */
int rmem_dwconv5x5_split2(const float *g0, const float *g1, int64_t ldg, const float *wt0, const float *wt1,
                          int32_t h, int32_t w, int32_t C, rmem_f16 *oh0, rmem_f16 *ol0, rmem_f16 *oh1,
                          rmem_f16 *ol1, int64_t ldo, void *stream);

/* Final GroupNorm1D(2 groups) over [tgt | tgt_id] (layers/transformer.py:806-808,
 * layers/basic.py:6-12): statistics over 256 channels x N tokens per group.
 * ws: >= 4*ceil(N/64) doubles. */
int rmem_groupnorm2(const float *tgt, const float *tgt_id, int32_t N, int32_t C,
                    const float *gamma, const float *beta, float eps, double *ws,
                    float *out, int64_t ldo, void *stream);
/* The same with the split-K partials of the LAST projection folded in first (transformer.py:1231-1232: tgt += o[:, :256],
 * tgt_id += o[:, 256:] of the self-attention projection): tgt[n][c] += sum_z parts[z*part_stride + n*ldpart + c],
 * tgt_id[n][c] += sum_z parts[.. + C + c], splits in order (rmem_layernorm_red's arithmetic); the statistics pass writes the
 * folded streams back, so tgt / tgt_id are inputs AND outputs.  Saves the launch that only folded. */
int rmem_groupnorm2_fold(float *tgt, float *tgt_id, const float *parts, int32_t nparts, int64_t part_stride, int64_t ldpart,
                         int32_t N, int32_t C, const float *gamma, const float *beta, float eps, double *ws, float *out,
                         int64_t ldo, void *stream);

/* ID assignment: label map -> one-hot(+ignore) -> Conv2d(k,stride,pad) -> LayerNorm_C
 * (utils/image.py:69-74, engines/aot_engine.py:208-232, models/aot.py:67-74,
 * models/deaot.py:65-69).  wt is [ncls][k][k][C] (k <= 31, C = 256); gamma NULL skips the LayerNorm (AOT).
 * ignore_channel != 0: label 255 selects the ignore channel ncls-1 (update_short_term_memory,
 * aot_engine.py:330-336, passes the real ignore mask); ignore_channel == 0: label 255 contributes
 * nothing (add_reference_frame, aot_engine.py:304, calls assign_identity without an ignore mask,
 * which :209-213 turn into all zeros). */
int rmem_id_assign(const uint8_t *label, int32_t H, int32_t W, const float *wt, const float *bias,
                   int32_t ncls, int32_t ksize, int32_t stride, int32_t pad, int32_t eh, int32_t ew,
                   int32_t C, const float *gamma, const float *beta, float eps,
                   rmem_f16 *oh, rmem_f16 *ol, int64_t ldo, float *of32, int64_t ldof,
                   int32_t ignore_channel, void *stream);

/* RMem relevance: out[t] = sum_q mass[q][t] * fg[q]   (layers/transformer.py:900-904) */
int rmem_attn_mass_reduce(const float *mass, int32_t N, int32_t T, const float *fg, float *out,
                          void *stream);

/* ---- RMem eviction on the device (restrict_long_memories, layers/transformer.py:880-991; trigger
 * engines/aot_engine.py:350-369) -- no device-to-host copy on the path.
 * rmem_fg_weights: fg[q] = 1 - softmax_c(bilinear, align_corners, of logits [C][Hl][Wl] at the h x w token grid)[0].
 * rmem_bank_state (device memory, one per clip): the per-slot dictionaries of the rule, aligned with the bank
 * positions of the logical->physical map `maps` (the int32 array rmem_read_args.slot_map points to).
 * rmem_bank_reset: the bank becomes the one slot `slot` (a reference frame); rmem_bank_append: the frame in `slot`
 * joins the bank; rmem_bank_policy_step: given w[t] = rmem_attn_mass_reduce(...) over the n_att attended slots,
 * EMA(0.8) + visit counts + UCB bonus + argmin, and -- when the bank holds more than cap slots -- the dropped
 * position is deleted from `maps` and the state.  result (may be NULL; pinned host or device memory) receives
 * {sequence number, dropped position or -1, slots left}, sequence number written last. */
typedef struct {
  int32_t T;                               /* slots in the bank                                   */
  int32_t index[16];                       /* frame index of bank position p (long_memories_indexes) */
  int32_t visits[16];                      /* stored_frame_times of position p (0 = not stored)   */
  int32_t has_ema[16];                     /* position p is in stored_attn_weight_dict            */
  float ema[16];
  int32_t last_drop;                       /* position dropped by the last step, -1 = none        */
  int32_t steps;                           /* policy steps taken so far                           */
} rmem_bank_state;

int rmem_fg_weights(const float *logits, int32_t C, int32_t Hl, int32_t Wl, int32_t h, int32_t w, float *fg,
                    void *stream);
int rmem_bank_reset(int32_t *maps, rmem_bank_state *st, int32_t slot, int32_t frame_index, void *stream);
int rmem_bank_append(int32_t *maps, rmem_bank_state *st, int32_t slot, int32_t frame_index, void *stream);
int rmem_bank_policy_step(int32_t *maps, rmem_bank_state *st, const float *w, int32_t n_att, int32_t cap,
                          int32_t former, int32_t *result, void *stream);

/* Support op outside the LSTT: GroupNorm(groups) (+ optional ReLU) on a contiguous NCHW
 * tensor of batch 1, as used by the FPN head's ConvGN blocks (decoders/fpn.py:43-62,
 * layers/basic.py:60-70).  Requires (C/groups)*HW % 4 == 0.  ws: >= 2*32*groups doubles. */
int rmem_groupnorm_nchw(const float *x, float *y, int32_t C, int64_t HW, int32_t groups,
                        const float *gamma, const float *beta, float eps, int32_t relu, double *ws,
                        void *stream);

/* Same, for the output x of a bias-free convolution: y = GN(x + conv_bias[c]) -- the ConvGN
 * block's nn.Conv2d bias (layers/basic.py:60-70) applied inside the statistics and the apply
 * pass instead of in a separate pass over the map. */
int rmem_groupnorm_nchw_bias(const float *x, const float *conv_bias, float *y, int32_t C, int64_t HW,
                             int32_t groups, const float *gamma, const float *beta, float eps,
                             int32_t relu, double *ws, void *stream);

/* Support op outside the LSTT: in-place conv epilogue y = act(x + bias[c] (+ residual)) on a
 * contiguous batch-1 NCHW tensor (the folded-FrozenBN bias, the bottleneck's residual add and
 * ReLU of encoders/resnet.py:47-69 in one pass). */
int rmem_bias_act_nchw(float *x, const float *bias, const float *residual, int32_t C, int64_t HW,
                       int32_t relu, void *stream);

/* Same for a contiguous [B][C][H][W] batch (bias indexed by channel): the encoder pass of several
 * announced frames at once (rmem_amd/engine.py, encoder prefetch). */
int rmem_bias_act_nchw_batched(float *x, const float *bias, const float *residual, int32_t B, int32_t C,
                               int64_t HW, int32_t relu, void *stream);

/* Support op outside the LSTT: y = (y + bias[c]) + bilinear_upsample(x -> H x W), in place on a
 * contiguous batch-1 NCHW map: the skip merge "adapter(shortcut) + F.interpolate(x)" of the FPN
 * head (decoders/fpn.py:53-60) in one pass.  bias may be NULL.  C*H <= 65535. */
int rmem_upsample_add_nchw(float *y, const float *bias, const float *x, int32_t C, int32_t H,
                           int32_t W, int32_t h, int32_t w, int32_t align_corners, void *stream);

/* Same, out of place: y_out = (y_in + bias[c]) + bilinear_upsample(x); y_in is left untouched (the
 * adapter's convolution output computed with the encoder pass may be decoded more than once). */
int rmem_upsample_add_nchw_out(const float *y_in, float *y_out, const float *bias, const float *x,
                               int32_t C, int32_t H, int32_t W, int32_t h, int32_t w,
                               int32_t align_corners, void *stream);

/* dst[0..n) = host_vals[0..n) (n <= 32), stream-ordered, payload in the kernel arguments: how
 * the logical->physical slot map of the bank is published without a blocking H2D copy. */
int rmem_set_ints(int32_t *dst, const int32_t *host_vals, int32_t n, void *stream);

/* fp32 -> planes (weights at load time, fixtures in tests) */
int rmem_split_planes(const float *x, int64_t n, rmem_f16 *hi, rmem_f16 *lo, void *stream);

/* Clip-driver post-processing.  One decoder output per test-time
 * augmentation: logits [C][h][w] fp32 (batch 1), flip != 0 when that augmentation ran on the
 * horizontally flipped frame. */
typedef struct {
  const float *logits;
  int32_t h, w;
  int32_t flip;
} rmem_label_src;

/* label[y][x] = argmax_c mean_a softmax_c(unflip_a(bilinear_a(logits_a -> H0 x W0)))
 * = F.interpolate(mode="bilinear", align_corners) of engines/aot_engine.py:457-463 followed by
 * flip_tensor / softmax / mean / argmax of managers/evaluator.py:424-441.  With one source the
 * softmax is skipped (monotone).  n_src <= 8, C <= 16; first maximum wins. */
int rmem_labels_from_logits(const rmem_label_src *srcs, int32_t n_src, int32_t C,
                            int32_t align_corners, int32_t H0, int32_t W0, uint8_t *label,
                            void *stream);

/* dst = F.interpolate(flip ? flip_x(src) : src, size=(Hd,Wd), mode="nearest") on uint8 label
 * maps (managers/evaluator.py:506-523, utils/image.py:107-111). */
int rmem_label_resize_nearest(const uint8_t *src, int32_t Hs, int32_t Ws, uint8_t *dst,
                              int32_t Hd, int32_t Wd, int32_t flip, void *stream);

/* ---- several clips' memory banks in one launch -------------------------------
 * The reference serves one clip per call: its attention asserts batch 1
 * (aot_plus/networks/layers/transformer.py:641,1190) and the evaluator walks clips one after the
 * other per process (managers/evaluator.py:344-523).  Here B clips of one geometry that advance in
 * lockstep share every launch of the memory path:
 *
 *   h = rmem_rec_begin();  ...any sequence of the calls below...;  rmem_rec_end(h);
 *
 * While a host thread is recording, rmem_linear / rmem_linear_grouped / rmem_layernorm_red[2] /
 * rmem_attn_read[2] / rmem_attn_read_combine[2] / rmem_dwconv5x5_split[2] /
 * rmem_groupnorm2 / rmem_id_assign / rmem_attn_mass_reduce on that thread validate their arguments
 * as usual but launch nothing: the argument block and launch geometry are appended to the
 * recording (other entry points are not recordable and launch immediately).  Recording the same
 * host code once per clip (different buffers) gives B argument blobs (rmem_rec_data/_size) with the
 * same rmem_rec_signature.  The caller places blob i at dev_args + i * clip_stride in device memory
 * (16-byte aligned, clip_stride >= blob size, multiple of 16) and
 *
 *   rmem_launch_recorded(h_of_any_clip, dev_args, clip_stride, B, stream)
 *
 * issues each recorded op ONCE for all B clips (grid z extent x B; a block reads its clip's
 * argument block from dev_args).  Per clip the arithmetic is the single-clip kernel's, bit for bit.
 * dev_args must stay unchanged until the launches have executed. */
void *rmem_rec_begin(void);                 /* NULL if this thread is already recording */
int rmem_rec_end(void *rec);
void rmem_rec_free(void *rec);
int32_t rmem_rec_count(const void *rec);    /* recorded ops */
int64_t rmem_rec_size(const void *rec);     /* bytes of the argument blob */
const void *rmem_rec_data(const void *rec);
uint64_t rmem_rec_signature(const void *rec);
int rmem_launch_recorded(const void *rec, const void *dev_args, int64_t clip_stride, int32_t B, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RMEM_HIP_H */
