/* swv2 -- C ABI of the MI355X-native SwinV2 weather hot path (libswv2.so).
 *
 * The reference (NERSC/swin_v2_weather, 100 % Python) has no FFI boundary: its hot path is the PyTorch op sequence
 * inside networks/swinv2_global.py.  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md): each entry point replaces the op sequence cited next to it.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, explicit sizes, no torch types.  `stream` is a hipStream_t passed as void*
 *    (0 = the null stream); every call only enqueues work on that stream and returns.
 *  - every function returns SWV2_OK (0) or a negative error code and never throws; `swv2_last_error()` returns a
 *    thread-local message for the last failure on the calling thread.
 *  - no entry point allocates, frees or synchronises: the caller owns all memory, including workspaces, so every
 *    call can be captured in a hipGraph.
 *  - no hidden global device state: re-entrant from autograd worker threads.
 *  - dtypes: activations that cross kernels are bf16 (uint16 storage) unless stated; the residual stream, LayerNorm
 *    statistics, log-sum-exp and all parameter gradients are fp32.  MFMA products take bf16 operands and
 *    accumulate in fp32.
 */
#ifndef SWV2_H
#define SWV2_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision: bumped whenever a struct layout or an entry point of this header changes (swv2_version() returns the value the
 * library was built with; swin_v2_weather_amd/_lib.py refuses a library whose revision differs from the one it mirrors).
 * 100 rounds 1 - 4; 105 round 5: swv2_attn_args.dbias_partials, swv2_block_desc.bias_prepacked / dbias_part, the *_multi CPB
 * entry points, swv2_attn_pack_bias_multi / swv2_attn_bias_chunks, a (max, min) part in the packed bias buffer;
 * 106: SWV2_EPI_UNPATCH_LOSS writes slot 1 of loss_part only where swv2_loss_part_reduce reads it; 107: swv2_epilogue.q[3], the loss
 * epilogue with the rollout destinations. */
#define SWV2_VERSION 107

enum {
    SWV2_OK = 0,
    SWV2_ERR_INVALID = -1,     /* bad argument (null pointer, size, alignment) */
    SWV2_ERR_UNSUPPORTED = -2, /* shape outside the compiled kernel set */
    SWV2_ERR_LAUNCH = -3       /* HIP launch failure */
};

int swv2_version(void);
const char* swv2_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Window-ordered, head-major attention layout (produced by swv2_linear with SWV2_EPI_QKV_HEADS):
 *   qkvh  [Bw][heads][3][Lp][DP] bf16   s=0: q/max(|q|,1e-12)  s=1: k/max(|k|,1e-12)  s=2: v ; zero padded
 *   rnorm [Bw][heads][2][Lp]     fp32   1/max(|q|,1e-12), 1/max(|k|,1e-12)
 *   oh    [Bw][heads][Lp][DP]    bf16   attention output, heads not yet merged
 *   lse   [Bw][heads][Lp]        fp32   log2-domain log-sum-exp of the scaled, biased, masked scores
 * Bw = B * windows per sample, window index = b*nW + wi*nww + wj, token index t = r*ww + c
 * (reference: window_partition, swinv2_global.py:89-101).  Lp, DP: swv2_attn_geometry().
 * ------------------------------------------------------------------------------------------------------------ */
int swv2_attn_geometry(int L, int head_dim, int* Lp, int* DP);

/* Which forward softmax regime serves window area L / head_dim (with / without a PACKED CPB bias table; dbg as in swv2_attn_args): 1 = the
 * operand-folded softmax of csrc/attn2.hip (160 .. 176-token windows; 16- and 32-wide head slots without a table, 16-wide ones with
 * a packed table; exponent reference sigma' (+ the table's maximum) where 2 sigma' + (table max - min) <= 80 in unmasked windows,
 * normaliser = sum of the bf16-rounded exponentials), 0 = row maximum + exact sum (csrc/attn.hip, attn_wide.hip); negative: unsupported geometry.
 * Pure host function.  The parity tests declare the regime their oracle emulates and check it against this (reference: the softmax of
 * swinv2_global.py:309-314, one function in both regimes up to rounding). */
int swv2_attn_fwd_regime(int L, int head_dim, int has_bias, int dbg);

#define SWV2_ATTN_FIRST_GEN 16
#define SWV2_ATTN_PLAIN_STATS 8192
#define SWV2_ATTN_BWD_TWO_PHASE 32

typedef struct swv2_attn_args {
    const void* qkvh;         /* in  */
    const float* logit_scale; /* in  [heads] raw tau; sigma = exp(min(tau, ln 100))   (swinv2_global.py:305) */
    const float* bias;        /* in  [heads][L][L] continuous position bias (swinv2_global.py:274-287) or NULL */
    const void* bias_pack;    /* optional: the same table pre-packed by swv2_attn_pack_bias (kernel register / LDS-image
                                 layouts, coalesced loads); NULL = the kernels convert `bias` themselves */
    void* oh;                 /* fwd: out; bwd: in */
    float* lse;               /* fwd: out; bwd: in */
    const void* doh;          /* bwd in : grad of oh, same layout */
    const float* rnorm;       /* bwd in */
    void* dqkvh;              /* bwd out: grads w.r.t. the un-normalised q, k, v, layout of qkvh */
    float* dlogit_scale;      /* bwd out: [heads], ACCUMULATED (caller zeroes) */
    float* dbias;             /* bwd out: [heads][L][L], ACCUMULATED (caller zeroes); NULL iff bias is NULL */
    int Bw, heads, L, head_dim;
    int nwh, nww;             /* windows per sample along H and W */
    int mask_thr;             /* shift mask (swinv2_global.py:403-424) in closed form: in the last window row
                                 (wi == nwh-1) tokens t >= mask_thr are region 1, others region 0; pairs from
                                 different regions get -100.  (wh - sh) * ww for a block shifted by sh > 0 rows;
                                 0 = no mask. */
    int max_chunks;           /* workgroups per head (each loops over windows); 64 is a good default (swv2_block_bwd uses
                                 256 / heads at the 176-token window: one persistent workgroup per CU) */
    int dbg;                  /* 0 in production.  Kernel-selection switches used by the parity tests (each selects a kernel that is
                                 also the product path of other shapes): SWV2_ATTN_FIRST_GEN = the first-generation kernels of
                                 csrc/attn.hip instead of the small-workgroup forward of csrc/attn2.hip; SWV2_ATTN_PLAIN_STATS =
                                 the two-phase backward with the softmax statistics read from LDS instead of riding in the
                                 MFMA operands; SWV2_ATTN_BWD_TWO_PHASE = the barrier-separated two-phase backward of csrc/attn.hip
                                 (statistics in the operands) where csrc/attn_bwd_stream.hip would run (176-row layout, 16-wide
                                 head slots, no table): same arithmetic, bit-identical d(qkv) */
    void* dbias_ws;           /* bwd, optional scratch of dbias_ws_bytes >= swv2_attn_dbias_ws_bytes(heads, L, max_chunks): the
                                 workgroups store their d bias tables there and one more launch sums them into dbias (in a
                                 fixed order); NULL / too small = 31 K float atomics per workgroup instead */
    size_t dbias_ws_bytes;
    int dbias_partials;       /* bwd, 1 (needs dbias_ws): LEAVE the workgroups' tables in dbias_ws -- [swv2_attn_bias_chunks(Bw)][heads][L][L],
                                 every entry written -- and do not touch dbias (may be NULL): the caller sums them, e.g. swv2_cpb_bwd_multi */
} swv2_attn_args;

/* bytes of swv2_attn_args.dbias_ws for swv2_attn_bwd with a bias */
size_t swv2_attn_dbias_ws_bytes(int heads, int L, int max_chunks);

/* workgroups per head of swv2_attn_bwd with a bias table (= tables per head in dbias_ws) when max_chunks is left to swv2_block_bwd */
int swv2_attn_bias_chunks(int Bw);

/* bias table -> the kernels' layouts (bf16, log2 domain) + the (max, min) of every head's packed values; out:
 * swv2_attn_pack_bias_bytes(heads, L) bytes.  _multi: ntab tables [ntab][heads][L][L] -> ntab packed buffers, one launch */
size_t swv2_attn_pack_bias_bytes(int heads, int L);
int swv2_attn_pack_bias(const float* bias, int heads, int L, void* out, void* stream);
int swv2_attn_pack_bias_multi(const float* bias, int ntab, int heads, int L, void* out, void* stream);

/* cosine window attention core, forward: swinv2_global.py:304-318 (q,k normalisation is in the QKV epilogue) */
int swv2_attn_fwd(const swv2_attn_args* a, void* stream);
/* backward of the same, including the backward of the L2 normalisation of q and k */
int swv2_attn_bwd(const swv2_attn_args* a, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Linear layers: C[M][N] = A[M][K] . W[N][K]^T on bf16 MFMA with fp32 accumulation.
 * The A operand is described by a "loader" (what PyTorch does as separate passes in the reference is folded into
 * the load), the output by an "epilogue".  W is always bf16 [N][K] row-major, produced by swv2_prep_weight.
 * Replaces: nn.Linear / nn.Conv2d(k=s=4) + the layout ops around them in swinv2_global.py (qkv :300-301, proj :319,
 * Mlp fc1/fc2 :381-386, PatchEmbed.proj :537,544, PatchMerging :519-523, head + un-patchify :767,784-792).
 * ------------------------------------------------------------------------------------------------------------ */
enum swv2_operand_kind {
    SWV2_OP_F32 = 0,       /* fp32 rows [rows][ld]; optional rowidx gather (roll + window partition, :457,89-101)   */
    SWV2_OP_BF16 = 1,      /* bf16 rows [rows][ld]; optional rowidx gather                                          */
    SWV2_OP_BF16_GELU = 2, /* bf16 rows, erf-GELU applied on load (timm Mlp act)                                     */
    SWV2_OP_HEADS = 3,     /* head-major window layout [Bw][h][S][Lp][DP]: row = bw*Lp + t, col = (part*h+head)*DP+j
                              p[0]=heads p[2]=Lp p[3]=DP, ld = S                                                     */
    SWV2_OP_PATCH = 4,     /* im2col of x[B][Cin][H][W] fp32 for the 4x4/stride-4 conv: row=(b,i,j), col=cin*16+p*4+q
                              p[0]=Cin p[1]=H p[2]=W ; p[3] = channels per sample of the tensor holding the Cin planes
                              (0 = Cin) ; aux0 = optional second source added on load, ld = its channels per sample      */
    SWV2_OP_MERGE_LN = 5,  /* 2x2 PatchMerging gather of x[B][H][W][C] fp32 + LayerNorm(4C) on load:
                              p[0]=H p[1]=W p[2]=C, aux0=mean aux1=rstd (swv2_merge_stats) aux2=gamma aux3=beta     */
    SWV2_OP_BF16_CSCALE = 6 /* bf16 rows [rows][ld] scaled on load by an fp32 factor per (sample, group of 16 columns):
                              value = bf16(ptr[row][col] * aux0[(row / p[0]) * p[2] + col / 16]) ; p[0] = rows per sample,
                              p[2] = groups per sample in aux0.  The loss-gradient operand of the head's backward: ptr = the
                              quadrature-weighted residual written by SWV2_EPI_UNPATCH_LOSS, aux0 = d loss / d S0 per
                              (sample, channel) (losses.py:188-206).  Accepted by swv2_linear with SWV2_EPI_F32 and as dY of
                              swv2_linear_wgrad* with an SWV2_OP_F32 X                                              */
};

typedef struct swv2_operand {
    int kind;              /* enum swv2_operand_kind */
    const void* ptr;
    const int32_t* rowidx; /* optional: logical row -> source row, negative = all-zero row */
    const float* aux0;
    const float* aux1;
    const float* aux2;
    const float* aux3;
    long ld;               /* row pitch in elements (or S for SWV2_OP_HEADS) */
    int rows, cols;        /* logical M and K */
    int p[4];
} swv2_operand;

enum swv2_epilogue_kind {
    SWV2_EPI_BF16 = 0,      /* out bf16 [M][ld] = acc + bias ; optional rowidx scatter                               */
    SWV2_EPI_F32 = 1,       /* out fp32 [M][ld] = acc + bias (+ aux fp32, same shape, added at the destination
                               row: the residual gradient) ; optional rowidx scatter                                 */
    SWV2_EPI_QKV_HEADS = 2, /* out = qkvh, aux_out = rnorm: + bias, split heads, L2-normalise q and k (:300-304)
                               p[0]=heads p[2]=Lp p[3]=DP p[4]=L ; N = 3*heads*DP, bias padded alike                 */
    SWV2_EPI_GELU_GRAD = 3, /* out bf16 = acc * GELU'(aux) ; aux = bf16 pre-activation, same shape/pitch as out     */
    SWV2_EPI_UNPATCH = 4,   /* out fp32 y[B][Cout][H][W] (+ aux skip[B][Cs][H][W]) ; N = Cout*16 with columns ordered
                               c*16+p*4+q ; p[0]=Cout p[1]=H p[2]=W p[3]=Cs (0 = no skip)   (:784-802) ;
                               p[4] = channels per sample of the tensor `out` points into (0 = Cout) ; aux_out = optional
                               second destination, ld = its channels per sample (rollout, helpers.py:26-41)          */
    SWV2_EPI_HEADS = 5,     /* out = [Bw][h][Lp][DP] bf16 split heads, no normalisation ; p as QKV_HEADS, N=heads*DP */
    SWV2_EPI_F32_ACC = 6,   /* out fp32 [M][ld] += acc ; optional rowidx scatter                                     */
    SWV2_EPI_BF16_GELU = 7, /* out bf16 = acc + bias (pre-activation, kept for backward), aux_out bf16 = erf-GELU of it,
                               both [M][ld] (fc1 of the timm Mlp, swinv2_global.py:381-386)                         */
    SWV2_EPI_UNPATCH_LOSS = 8 /* SWV2_EPI_UNPATCH (single destination) that also evaluates the geometric l2 loss sums of
                               the prediction it is writing (losses.py:188-206, grids.py:115-117) against the target
                               loss_tar fp32 [B][q[0]][H][W] (channels q[1] .. q[1] + Cout): per (sample, channel)
                               sum_hw qw[h] (y - tar)^2 and sum_hw qw[h] tar^2, left as per-row-group partial sums
                               loss_part[g][slot][c][2] (g = group of SWV2_LOSS_GROUP_ROWS rows of the GEMM; slot 0 = rows of the sample of the
                               group's first row, slot 1 = rows of the following sample -- WRITTEN only by groups whose successor
                               row belongs to another sample or lies past M, the only groups swv2_loss_part_reduce reads it of;
                               zero there unless the group straddles the boundary; plain stores, no atomics: same-address float atomics from every workgroup
                               measured 8 x the whole kernel) which swv2_loss_part_reduce folds in a fixed order.  Also
                               writes loss_resid bf16 [M][SWV2_LOSS_RESID_PITCH(N)] = qw[h] (y - tar) in the GEMM's own row / column order -- the
                               operand SWV2_OP_BF16_CSCALE feeds to the head's backward, so neither the prediction nor a
                               materialised gradient is read again.  A operand: SWV2_OP_F32 only; at least
                               SWV2_LOSS_GROUP_ROWS rows per sample; every tensor below 2^32 elements; `out` and loss_resid must be
                               followed by SWV2_LOSS_DUMP_BYTES of write-only scratch (masked lanes store there, so that the
                               epilogue is branch-free).  Rollouts (107): p[4] = channels per sample of the tensor `out` points into
                               (with q[2]), aux_out / ld = a second destination [B][ld][H][W] for the same prediction (the next
                               step's input, also followed by SWV2_LOSS_DUMP_BYTES), as SWV2_EPI_UNPATCH has them  */
};

typedef struct swv2_epilogue {
    int kind;              /* enum swv2_epilogue_kind */
    void* out;
    const float* bias;     /* [N] or NULL */
    const void* aux;
    float* aux_out;
    const int32_t* rowidx; /* optional: logical row -> destination row, negative = skip */
    long ld;               /* output row pitch in elements */
    int p[5];
    /* SWV2_EPI_UNPATCH_LOSS only (ignored by every other kind) */
    const float* loss_tar;   /* target [B][q[0]][H][W] fp32 */
    const float* loss_qw;    /* quadrature row weights [H] */
    float* loss_part;        /* [ceil(M / SWV2_LOSS_GROUP_ROWS)][2][Cout][2] fp32 per-group partial sums: slot 0 overwritten, slot 1 see above */
    void* loss_resid;        /* bf16 [M][SWV2_LOSS_RESID_PITCH(N)], columns 0 .. N - 1 written */
    int q[3];                /* channels per sample of loss_tar, first target channel of this prediction, and (rollouts: p[4] != 0, `out`
                                points into a [B][p[4]][H][W] tensor) the offset in floats from `out` to that tensor's dump area; 0 = behind
                                the dense [B][Cout][H][W] prediction */
} swv2_epilogue;

int swv2_linear(const swv2_operand* a, const void* w_bf16, const swv2_epilogue* e, int N, void* stream);

/* Weight gradient: dW[nmap(n)][kmap(k)] += sum_m dY[m][n] X[m][k], db[nmap(n)] += sum_m dY[m][n]  (caller zeroes; db
 * may be NULL).  dW has row pitch ldw; nmap/kmap (NULL = identity, negative = drop) undo the head padding of the
 * SWV2_OP_HEADS layouts.  splits > 0 = number of row slices processed by different workgroups (x 2 for outputs of fewer
 * than three 128 x 128 tiles).  swv2_linear_wgrad sums the slices with fp32 atomics on dW; swv2_linear_wgrad_ws writes
 * per-slice partial tiles to the caller's workspace (>= swv2_linear_wgrad_ws_bytes(M, N, K, splits) bytes, contents
 * undefined afterwards) and adds them in a second kernel -- no atomics on dW, deterministic, and faster from ~64
 * slices up.  ws == NULL falls back to the atomic path.  Replaces the autograd of F.linear / Conv2d weights. */
int swv2_linear_wgrad(const swv2_operand* dy, const swv2_operand* x, float* dW, float* db, const int32_t* nmap,
                      const int32_t* kmap, int ldw, int splits, void* stream);
size_t swv2_linear_wgrad_ws_bytes(int M, int N, int K, int splits);
int swv2_linear_wgrad_ws(const swv2_operand* dy, const swv2_operand* x, float* dW, float* db, const int32_t* nmap,
                         const int32_t* kmap, int ldw, int splits, void* ws, size_t ws_bytes, void* stream);

/* The four weight gradients of one transformer block in ONE launch + one reduction (item 0: fc2 = (BF16, BF16_GELU),
 * 1: fc1 = (BF16, F32), 2: proj = (BF16, HEADS), 3: qkv = (HEADS, F32); any other operand kinds -> SWV2_ERR_INVALID).
 * Same results as four swv2_linear_wgrad_ws calls up to the summation order of the row slices.  slices = 0 picks the
 * count that fills the chip in one round (2 workgroups per CU); ws >= swv2_block_wgrad_ws_bytes(C, hidden, heads * DP,
 * slices) bytes.  Replaces: the autograd of the block's four nn.Linear weights (swinv2_global.py:146-201, 231-248). */
typedef struct swv2_wgrad_item {
    swv2_operand dy, x;
    float* dW;
    float* db;             /* optional */
    const int32_t* nmap;
    const int32_t* kmap;
    int ldw;
} swv2_wgrad_item;
size_t swv2_block_wgrad_ws_bytes(int C, int hidden, int heads_dp, int slices);
int swv2_block_wgrad(const swv2_wgrad_item* items4, int slices, void* ws, size_t ws_bytes, void* stream);

/* Head dims 65 .. 128 (DP = 96 up to 96 channels, else 128): SWV2_EPI_QKV_HEADS then leaves the squared norms of q, k in rnorm (which the
 * caller zeroes first) and un-normalised values in qkvh; this pass finishes F.normalize (swinv2_global.py:300-304):
 * rnorm <- 1 / max(|.|, 1e-12) (0 on rows >= L) and q, k rescaled in place. */
int swv2_qk_normalize(void* qkvh_bf16, float* rnorm, int Bw, int heads, int Lp, int L, int DP, void* stream);

/* out_bf16[i][j] = W'[row_map ? row_map[i] : i][col_map ? col_map[j] : j] (0 where a map entry is negative),
 * W' = transpose ? w^T : w, w fp32 [rows][cols].  Casts, transposes, permutes and pads a parameter once per step. */
int swv2_prep_weight(const float* w, int rows, int cols, int transpose, const int32_t* row_map, int out_rows,
                     const int32_t* col_map, int out_cols, void* out_bf16, void* stream);

/* The same for many parameters in ONE launch (all prepared copies of a model after an optimizer step).  items_dev: device
 * array; chunks_dev: device array of int pairs (item index, chunk index within the item's output), one workgroup per chunk
 * of swv2_prep_chunk() output elements; both tables are built by the caller (swin_v2_weather_amd/ops.py::PrepBatch).
 * out_f32 != 0: the output stays fp32 (the head-padded qkv bias). */
typedef struct swv2_prep_item {
    const float* w;
    int rows, cols, transpose;
    const int32_t* row_map;
    int out_rows;
    const int32_t* col_map;
    int out_cols;
    void* out;
    int out_f32;
} swv2_prep_item;
int swv2_prep_chunk(void);
/* number of workgroups (chunk indices 0 .. n - 1 of the pair table) item needs: linear chunks of swv2_prep_chunk() outputs, or
 * 64 x 64 output tiles for transposed copies of matrices with both dimensions >= 64 (turned in LDS) */
int swv2_prep_item_chunks(int out_rows, int out_cols, int transpose);
int swv2_prep_multi(const swv2_prep_item* items_dev, const int* chunks_dev, int n_chunks, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused LayerNorm + drop-path + residual add, with the window_reverse + reverse cyclic roll folded into the row
 * scatter:  y[dst] = res[res_mod ? dst % res_mod : dst] + scale[dst / rows_per_sample] * (LN(a[m])*gamma + beta),
 * dst = rowidx ? rowidx[m] : m (negative = padded row, skipped).
 * Replaces  x + drop_path(norm(branch))  (swinv2_global.py:490,496), window_reverse + roll (:468-476) and
 * PatchEmbed's norm + pos_embed add (:545,780 with res = pos_embed as [T][C], res_mod = T).
 * ------------------------------------------------------------------------------------------------------------ */
#define SWV2_LN_BWD_MAX_BLOCKS 512
typedef struct swv2_ln_args {
    const void* a;         /* bf16 [M][C] branch output                                   */
    const float* res;      /* fwd: fp32 residual rows (NULL = 0)                           */
    const float* gamma;    /* [C] */
    const float* beta;     /* [C] (fwd) */
    const float* scale;    /* [B] drop-path factor per sample or NULL                     */
    const int32_t* rowidx; /* [M] or NULL                                                  */
    float* y;              /* fwd out: fp32 rows                                           */
    float* mean;           /* [M] fwd: out, bwd: in */
    float* rstd;           /* [M] fwd: out, bwd: in */
    const float* dy;       /* bwd in : fp32 grad of y (destination rows)                  */
    void* da;              /* bwd out: bf16 [M][C] grad of a (zeros on padded rows)       */
    float* dgamma;         /* bwd out: [C] ACCUMULATED */
    float* dbeta;          /* bwd out: [C] ACCUMULATED */
    float* ws;             /* bwd: workspace of SWV2_LN_BWD_MAX_BLOCKS * 2 * C floats (per-block partial sums) */
    int M, C, res_mod, rows_per_sample;
    float eps;
} swv2_ln_args;

int swv2_ln_residual_fwd(const swv2_ln_args* a, void* stream);
int swv2_ln_residual_bwd(const swv2_ln_args* a, void* stream);

/* out[i] (+)= sum_b in[b][i], i < n (n % 4 == 0): pos_embed gradient */
int swv2_batch_sum(const float* in, float* out, int B, long n, int accumulate, void* stream);

/* PatchMerging (swinv2_global.py:500-523; never instantiated by swinv2net but part of the model file) */
int swv2_merge_stats(const float* x, float* mean, float* rstd, int B, int H, int W, int C, float eps, void* stream);
int swv2_merge_ln_bwd(const float* x, const void* dn_bf16, const float* gamma, const float* mean, const float* rstd,
                      float* dx, float* dgamma, float* dbeta, int B, int H, int W, int C, void* stream);

/* Geometric l2 loss (utils/losses.py:188-232, utils/grids.py:115-117): one pass produces, per (b,c) plane,
 * sums[2*bc] += sum q[h](prd-tar)^2 and sums[2*bc+1] += sum q[h] tar^2 (caller zeroes sums);
 * the backward is dprd = coef[bc] * q[h] * (prd - tar) with coef from the chain rule of the chosen variant. */
int swv2_loss_sums(const float* prd, const float* tar, const float* quad_w, float* sums, int BC, int H, int W, void* stream);
int swv2_loss_grad(const float* prd, const float* tar, const float* quad_w, const float* coef, float* dprd, int BC, int H,
                   int W, void* stream);
/* The scalar on top of the sums (losses.py:200-232): loss = sum_{b,c} chw[c] f(S0 / S1) (S0 alone when absolute; f = sqrt
 * unless squared) and coef[bc] = 2 d loss / d S0[bc], the factor swv2_loss_grad takes.  chw: [C] weights (channel weights x
 * multistep weights, normalised as LossHandler does); BC = B * C. */
/* sums is [layers][BC][2]: the layers are added in order (layers = 1: swv2_loss_sums; SWV2_LOSS_PART_SLICES:
 * swv2_loss_part_reduce). */
#define SWV2_LOSS_PART_SLICES 8
#define SWV2_LOSS_GROUP_ROWS 32
#define SWV2_LOSS_DUMP_BYTES 2048
/* row pitch (elements) of loss_resid for N = Cout * 16 columns: whole 128-byte lines per row, so that the 64-column pieces the epilogue
 * writes and the head's backward products read never straddle a line (107) */
#define SWV2_LOSS_RESID_PITCH(N) (((N) + 63) / 64 * 64)
/* Folds the loss epilogue's per-group partial sums: sums[j][b][coff + c][k] = sum over the SWV2_LOSS_GROUP_ROWS-row groups g with
 * g % SWV2_LOSS_PART_SLICES == j of (slot 0 of g if g's first row lies in sample b) + (slot 1 of g if it lies in sample b - 1),
 * ascending g; T rows per sample (T >= SWV2_LOSS_GROUP_ROWS).  sums: [SWV2_LOSS_PART_SLICES][B][Ct][2], overwritten for the Cout channels from coff. */
int swv2_loss_part_reduce(const float* part, int M, int T, int B, int Cout, int Ct, int coff, float* sums, void* stream);
/* The loss epilogue's residual (loss_resid, bf16 [M][SWV2_LOSS_RESID_PITCH(Cout * 16)], rows = patches (b, i, j), columns c * 16 + p * 4 + q) back in image layout and
 * scaled: out[b][c][4i + p][4j + q] = coef[b][c] * resid (+ add[b][c][..] when add != NULL; add: [B][Cadd][H][W], out: the first Cout channels of
 * [B][Cs][H][W]) -- d loss / d prediction where a consumer wants it as an image (the skip connection of a rollout step,
 * swinv2_global.py:799-801).  coef as for swv2_loss_grad.  (107) */
int swv2_loss_resid_to_image(const void* resid, const float* coef, const float* add, float* out, int B, int Cout, int H, int W, int Cs, int Cadd,
                             void* stream);
int swv2_loss_finalize(const float* sums, int layers, const float* chw, int BC, int C, int absolute, int squared, float* loss, float* coef,
                       void* stream);

/* torch.optim.Adam step (train.py:176) over one flat fp32 buffer; step >= 1; grads are multiplied by grad_inv_scale */
int swv2_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   int step, float grad_inv_scale, void* stream);

/* The same update for MANY tensors in one launch (replaces optimizer.step() of train.py:176 / :330).  items_dev: device
 * array of tensors; chunks_dev: device array of int pairs (item index, chunk index within the item), one workgroup per
 * chunk of swv2_adam_chunk() elements; both tables are built by the caller (swin_v2_weather_amd/utils/optim.py). */
typedef struct swv2_adam_item {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
} swv2_adam_item;
int swv2_adam_chunk(void);
int swv2_adam_multi(const swv2_adam_item* items_dev, const int* chunks_dev, int n_chunks, float lr, float beta1, float beta2,
                    float eps, int step, float grad_inv_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * ERA5 input assembly (the step BEFORE the model): raw time slabs staged on the device by async H2D copies
 * -> the model's input / target buffers, one pass each.
 *   raw  [B][S][Craw][Hraw][Wraw] fp32   S time slabs per sample as stored in the year files (721 x 1440 rows, uncropped)
 *   out  [B][Cout_total][H][W]    fp32   channels coff .. coff + S*Csel - 1 are written:
 *        out[b][coff + s*Csel + c][i][j] = (raw[b][s][chan[c]][i][j] - mean[c]) / std[c]        i < H, j < W
 * replaces the host-side crop + channel select + z-score of utils/data_loader_era5.py:163-171,98-107 (bit-identical: same
 * operation order, IEEE division) and DALI's fn.normalize (utils/data_loader_era5_dali.py:77-90).
 * swv2_era5_zenith writes nz cos-zenith channels, the per-pixel half of data_loader_era5.py:109-146 (modulus cos_zenith_angle):
 * out[b][coff + k][i][j] = sin(lat_i) sun0 + cos(lat_i) sun1 cos(sun2 + lon_j) on the 0.25 degree grid (lat 90 .. -90, lon 0 ..),
 * sun[(b*nz + k)*3 + {0, 1, 2}] = sin(declination), cos(declination), hour angle at longitude 0 of the time point -- the solar
 * position is float64 host arithmetic (utils/data_loader_era5.py::sun_position; oracle/zenith.py restates the routine);
 * swv2_era5_static broadcasts Cs static feature planes [Cs][H][W] over the batch (utils/preprocess_utils.py:50-68).
 * ------------------------------------------------------------------------------------------------------------ */
int swv2_era5_select_normalize(const float* raw, float* out, const int* chan, const float* mean, const float* stdv, int B, int S,
                               int Csel, int Craw, int Hraw, int Wraw, int H, int W, int Cout_total, int coff, void* stream);
int swv2_era5_zenith(float* out, const float* sun, int B, int nz, int H, int W, int Cout_total, int coff, void* stream);
int swv2_era5_static(const float* stat, float* out, int B, int Cs, int H, int W, int Cout_total, int coff, void* stream);

/* Output projection of the attention branch fused with LayerNorm1 (swinv2_global.py:318-319, 468-476, 490):
 *   forward : y[dst] = x[dst] + scale[b] * LN(merge_heads(oh) Wp^T + bp),  dst = rowidx[m] (negative = padded row, skipped);
 *             saves a1 = bf16(proj output) [Bw*Lp][C] (window order), mean, rstd            (replaces swv2_linear + swv2_ln_residual_fwd)
 *   backward: da1 = LN backward of scale * dy[dst] (zeros for padded rows), d(oh) = split_heads(da1 Wp), head-major;
 *             dgamma / dbeta ACCUMULATED; ws >= swv2_proj_ln_bwd_ws_floats(Bw*Lp, C) floats   (replaces swv2_ln_residual_bwd + swv2_linear)
 * C in {32,64,96,128}, head dim padded to 16, an even number of heads with heads * 16 <= 128; or C = 192 with up to 8 heads in
 * 32-wide slots (oh / doh [Bw][heads][Lp][32], wp [192][heads*32], wpt [heads*32][192]: BASELINE configs[4])  (swv2_proj_ln_supported). */
typedef struct {
    const void* oh;        /* bf16 [Bw][heads][Lp][16] */
    const void* wp;        /* bf16 [C][heads*16] (swv2_prep_weight with the head-padding column map) */
    const float* bp;       /* [C] */
    const float* gamma; const float* beta; const float* scale;
    const int32_t* rowidx; /* [Bw*Lp] or NULL (identity) */
    const float* x;        /* fp32 [rows][C] residual, destination order */
    void* a1; float* mean; float* rstd;     /* out, window order */
    float* y;              /* out fp32 [rows][C], destination order */
    int Bw, Lp, heads, C, rows_per_sample;
    float eps;
} swv2_proj_ln_args;
typedef struct {
    const float* dy;       /* fp32 [rows][C] gradient of y, destination order */
    const void* a1; const float* mean; const float* rstd; const float* gamma; const float* scale;
    const int32_t* rowidx;
    const void* wpt;       /* bf16 [heads*16][C] = proj.weight^T (swv2_prep_weight, transpose, head-padding row map) */
    void* da1;             /* out bf16 [Bw*Lp][C] */
    void* doh;             /* out bf16 [Bw][heads][Lp][16] */
    float* dgamma; float* dbeta; float* ws;
    int Bw, Lp, heads, C, rows_per_sample;
} swv2_proj_ln_bwd_args;
int swv2_proj_ln_supported(int C, int heads, int head_pad);
size_t swv2_proj_ln_bwd_ws_floats(int Mw, int C);
int swv2_proj_ln_fwd(const swv2_proj_ln_args* a, void* stream);
int swv2_proj_ln_bwd(const swv2_proj_ln_bwd_args* a, void* stream);

/* Fused MLP branch, forward: y = x + scale[b] * LayerNorm(fc2(GELU(fc1(x))))  (swinv2_global.py:492-496, timm Mlp
 * :381-386) in one kernel; the [M][hidden] activation stays in registers.  Saves for the backward: hpre = bf16(fc1(x)),
 * a2 = bf16(fc2 output), mean / rstd of the LayerNorm.  Same results as swv2_linear(EPI_BF16_GELU) + swv2_linear +
 * swv2_ln_residual_fwd (identical rounding points).  C in {32,64,96,128,192,256}, hidden % 32 == 0,
 * hidden <= 2048 (swv2_mlp_supported); other shapes return SWV2_ERR_UNSUPPORTED -- use the unfused sequence. */
typedef struct {
    const float* x;        /* [M][C] fp32 rows: GEMM input and residual */
    const void* w1;        /* bf16 [hidden][C]  (swv2_prep_weight) */
    const float* b1;       /* [hidden] */
    const void* w2;        /* bf16 [C][hidden] */
    const float* b2;       /* [C] */
    const float* gamma;    /* LayerNorm weight / bias [C] */
    const float* beta;
    const float* scale;    /* per-sample drop-path factor [M / rows_per_sample] or NULL */
    void* hpre;            /* out bf16 [M][hidden] */
    void* a2;              /* out bf16 [M][C] */
    float* mean;           /* out [M] */
    float* rstd;
    float* y;              /* out fp32 [M][C] */
    int M, C, hidden, rows_per_sample;
    float eps;
} swv2_mlp_args;
int swv2_mlp_supported(int C, int hidden);
int swv2_mlp_recompute_supported(int C, int hidden);      /* the backward's recompute mode (swv2_mlp_bwd_args.hpre == NULL) */
int swv2_mlp_fwd(const swv2_mlp_args* a, void* stream);

/* Fused MLP branch, backward (the autograd of swv2_mlp_fwd w.r.t. x and the data-path intermediates):
 *   da2 = LayerNorm backward of scale[b] * dy (saved a2, mean, rstd);  dh = (da2 W2) * GELU'(hpre);  dx = dy + dh W1
 * in one kernel (replaces swv2_ln_residual_bwd + swv2_linear(EPI_GELU_GRAD) + swv2_linear(EPI_F32)).  da2 and dh are
 * written for the weight gradients (swv2_linear_wgrad: d fc2 = da2^T GELU(hpre), d fc1 = dh^T x).  dgamma / dbeta are
 * ACCUMULATED.  ws: >= swv2_mlp_bwd_ws_floats(M, C) floats, contents undefined afterwards. */
typedef struct {
    const float* dy;       /* [M][C] gradient of y */
    const void* a2;        /* bf16 [M][C]      saved by the forward */
    const float* mean;     /* [M] */
    const float* rstd;
    const float* gamma;    /* [C] */
    const float* scale;    /* per-sample drop-path factor or NULL */
    const void* hpre;      /* bf16 [M][hidden] saved by the forward, or NULL (recompute mode, below) */
    const void* w2t;       /* bf16 [hidden][C] = fc2.weight^T (swv2_prep_weight, transpose) */
    const void* w1t;       /* bf16 [C][hidden] = fc1.weight^T */
    void* da2;             /* out bf16 [M][C] */
    void* dh;              /* out bf16 [M][hidden] */
    float* dx;             /* out fp32 [M][C] */
    float* dgamma;         /* [C], accumulated */
    float* dbeta;
    float* ws;
    int M, C, hidden, rows_per_sample;
    /* recompute mode (hpre == NULL; w1t unused): the pre-activation is rebuilt from the forward's input instead of read back
     * -- the forward (swv2_mlp_args.hpre == NULL) then never writes it: 16 bytes per token-channel less per block */
    const float* x;        /* [M][C] the forward's input */
    const void* w1;        /* bf16 [hidden][C] fc1.weight */
    const float* b1;       /* [hidden] */
} swv2_mlp_bwd_args;
size_t swv2_mlp_bwd_ws_floats(int M, int C);
int swv2_mlp_bwd(const swv2_mlp_bwd_args* a, void* stream);

/* Continuous position bias (swinv2_global.py:240-261,274-287): bias[heads][L][L] = meta_mlp(log-spaced relative
 * coordinates), meta_mlp = Linear(2,hidden) -> ReLU -> Dropout(drop_p) -> Linear(hidden,heads); the relative-coordinate
 * table is generated in-kernel.  keep_bf16: [L*L][hidden] keep-mask drawn by the caller (any non-zero = keep; the kernel
 * applies 1/(1-drop_p)), NULL in eval mode.  Backward outputs are ACCUMULATED (caller zeroes). */
int swv2_cpb_fwd(const float* w1, const float* b1, const float* w2, const float* b2, const void* keep_bf16, float* bias,
                 int wh, int ww, int heads, int hidden, float drop_p, void* stream);
int swv2_cpb_bwd(const float* dbias, const float* w1, const float* b1, const float* w2, const void* keep_bf16, float* dw1,
                 float* db1, float* dw2, float* db2, int wh, int ww, int heads, int hidden, float drop_p, void* stream);
/* The same with a caller workspace of swv2_cpb_bwd_ws_bytes(wh, ww, heads, hidden) bytes: every workgroup leaves one partial row,
 * a second launch adds them in a fixed order -- no float atomics (bit-reproducible gradients).  ws = NULL or too small: the atomics
 * path of swv2_cpb_bwd.  Outputs ACCUMULATED as above. */
size_t swv2_cpb_bwd_ws_bytes(int wh, int ww, int heads, int hidden);
int swv2_cpb_bwd_ws(const float* dbias, const float* w1, const float* b1, const float* w2, const void* keep_bf16, float* dw1,
                    float* db1, float* dw2, float* db2, int wh, int ww, int heads, int hidden, float drop_p, void* ws,
                    size_t ws_bytes, void* stream);

/* The tables of ALL blocks of a stage in one launch each way (nothing in :240-261,274-287 depends on activations, so the model
 * computes the `nblk` tables before its first block and their parameter gradients after the first block's backward).
 *   params_dev : DEVICE array [nblk][4] of device pointers: w1 [hidden][2], b1 [hidden], w2 [heads][hidden], b2 [heads] of each block
 *   keep_bits  : u32 [nblk][L*L][hidden / 8] of uniformly random bits drawn by the caller (31 random bits per word are enough: only the
 *                three low bytes are used), or NULL (eval).  Hidden unit j of a pair <-> bit (j & 7) of W | W >> 8 | W >> 16 with
 *                W = word ((j >> 3) & 3) * (hidden / 32) + (j >> 5) of the pair: the unit is DROPPED iff that bit is clear in all three
 *                low bytes of W -- probability 1/8, the reference's hard-coded Dropout(0.125) (:245; drop_p must be 0.125); kept units
 *                are scaled by 1 / (1 - drop_p)
 *   bias       : out [nblk][heads][L][L]
 * backward: dbias_tables [nblk][nchunk][heads][L][L] -- d bias of a block = the SUM of its nchunk tables (the per-workgroup tables
 * swv2_attn_bwd leaves with dbias_partials; nchunk = 1: plain gradients) -- grads [nblk][3 hidden + heads hidden + heads] =
 * (dw1 | db1 | dw2 | db2) per block, ACCUMULATED; ws: swv2_cpb_bwd_multi_ws_bytes.  heads <= 16, hidden in {64, 128, 256, 384, 512}.  No atomics;
 * both directions run their contractions on the matrix pipe with hi + lo split bf16 operands (fp32 accuracy: ~1e-5 relative). */
int swv2_cpb_fwd_multi(const float* const* params_dev, int nblk, const uint32_t* keep_bits, float* bias, int wh, int ww, int heads,
                       int hidden, float drop_p, void* stream);
size_t swv2_cpb_bwd_multi_ws_bytes(int nblk, int wh, int ww, int heads, int hidden);
int swv2_cpb_bwd_multi(const float* dbias_tables, int nchunk, const float* const* params_dev, int nblk, const uint32_t* keep_bits,
                       float* grads, int wh, int ww, int heads, int hidden, float drop_p, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Whole-block orchestration: one host call enqueues the 7 forward / 13 backward launches of a Swin block
 * (reference SwinTransformerV2CrBlock.forward, swinv2_global.py:480-497, and its autograd) from C++, so the per-launch
 * host cost is a few microseconds instead of a Python round trip.  All memory is caller-owned.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct swv2_block_desc {
    /* geometry */
    int B, T, C, heads, head_dim, hidden, L, Lp, DP, nwh, nww, mask_thr;
    const int32_t* rowidx;   /* [Bw*Lp] window-ordered padded row -> image row (b*T + i*gw + j) or -1 */
    const int32_t* qkv_map;  /* [3*heads*DP] padded qkv feature -> qkv feature or -1 */
    const int32_t* proj_map; /* [heads*DP]   padded head feature -> channel or -1 */
    /* parameters (fp32) and their prepared bf16 copies (swv2_prep_weight) */
    const float *logit_scale, *qkv_b_pad, *proj_b, *n1_w, *n1_b, *fc1_b, *fc2_b, *n2_w, *n2_b;
    const void *w_qkv, *w_proj, *w_fc1, *w_fc2;     /* forward: [3hDP][C], [C][hDP], [hid][C], [C][hid] */
    const void *w_qkvt, *w_projt, *w_fc1t, *w_fc2t; /* backward (transposed): [C][3hDP], [hDP][C], [C][hid], [hid][C] */
    /* per-call inputs */
    const float* x;          /* [B*T][C] block input (residual stream) */
    const float* bias;       /* [heads][L][L] CPB table or NULL */
    void* bias_pack;         /* with bias: swv2_attn_pack_bias_bytes(heads, L) bytes, written by the forward, read by both */
    const float* dp1;        /* [B] drop-path scales or NULL */
    const float* dp2;
    /* saved activations: written by forward, read by backward */
    void* qkvh;  float* rnorm; void* oh; float* lse; void* a1; float* mean1; float* rstd1; float* x1;
    void* hpre;  void* hact;   void* a2; float* mean2; float* rstd2;
    float* x2;               /* forward output [B*T][C] */
    /* backward only */
    const float* dx2;        /* grad of x2 */
    void *da2, *dh, *da1, *doh, *dqkvh;   /* bf16 scratch: [BT][C], [BT][hid], [Bw*Lp][C], [Bw][h][Lp][DP], [Bw][h][3][Lp][DP] */
    float* dx1;              /* fp32 scratch [BT][C] */
    float* ln_ws;            /* max(SWV2_LN_BWD_MAX_BLOCKS*2*C, swv2_mlp_bwd_ws_floats(B*T, C), swv2_proj_ln_bwd_ws_floats(Bw*Lp, C)) floats */
    float* dx;               /* out: grad of x */
    /* parameter gradients, ACCUMULATED (caller zeroes) */
    float *d_logit_scale, *d_bias, *d_qkv_w, *d_qkv_b, *d_proj_w, *d_proj_b, *d_n1_w, *d_n1_b, *d_fc1_w, *d_fc1_b,
          *d_fc2_w, *d_fc2_b, *d_n2_w, *d_n2_b;
    int wgrad_splits;        /* row slices of the weight-gradient products (128 with a workspace, 64 without) */
    /* optional timing of ONE launch with HIP events on the launch stream (bench.py's roofline): if ev_kernel matches a
       launch id (forward 1 qkv, 2 attn_fwd, 3 proj, 4 ln1, 5 fc1, 6 fc2, 7 ln2; backward 11 ln2, 12 wgrad fc2, 13 dh,
       14 wgrad fc1, 15 dx1, 16 ln1, 17 wgrad proj, 18 d(oh), 19 attn_bwd, 20 wgrad qkv, 21 dx), ev_start / ev_stop
       (hipEvent_t) are recorded around it; 0 = off */
    int ev_kernel;
    void* ev_start;
    void* ev_stop;
    int fuse_proj_ln;        /* 1: proj + LN1 run as swv2_proj_ln_fwd / _bwd when the shape is supported (forward steps 3-4,
                                backward steps 16, 18) */
    int fuse_attn;           /* reserved, must be 0 (the one-kernel attention branch of round 2 did not beat the four kernels and
                                lives in tools/experiments/attn_fused.hip, outside the library) */
    int fuse_mlp;            /* 1: forward steps 5-7 run as swv2_mlp_fwd when the shape is supported (hact is then neither
                                written nor read: the backward applies GELU to hpre on load); 0: three launches.  The backward
                                likewise runs steps 11, 13, 15 as swv2_mlp_bwd */
    void* wgrad_ws;          /* optional workspace of the weight-gradient products (swv2_linear_wgrad_ws), shared by the four
                                products of the block (they are ordered on one stream); NULL = atomic accumulation */
    size_t wgrad_ws_bytes;
    int wgrad_side_stream;   /* 1: backward launches the 4 weight-gradient products on the library's per-device side stream
                                (fork after each producer, join before returning) so they overlap with the dX chain */
    int wgrad_group;         /* 1 (with fuse_mlp + fuse_proj_ln paths and a workspace): the four products run as ONE
                                swv2_block_wgrad launch at the end of the backward (launch id 22) */
    void* grad_zero;         /* optional: one buffer holding all 13 parameter gradients of the block (16-byte aligned, a multiple
                                of 16 bytes); with the fused MLP path the backward's first kernel zeroes it, so the caller need
                                not (NULL: the caller zeroes every d_* buffer itself) */
    size_t grad_zero_bytes;
    size_t ln_ws_floats;     /* floats available at ln_ws; >= swv2_mlp_bwd_ws_floats + swv2_proj_ln_bwd_ws_floats lets the backward
                                keep both LayerNorms' d gamma / d beta partial rows and fold them with ONE launch (0: one each) */
    int bias_prepacked;      /* 1: bias_pack already holds the packed table (swv2_attn_pack_bias_multi: all blocks' tables in one
                                launch before the first block); 0: swv2_block_fwd packs `bias` into bias_pack itself */
    float* dbias_part;       /* optional, with bias: swv2_attn_dbias_ws_bytes(heads, L, swv2_attn_bias_chunks(Bw)) bytes.  The backward
                                LEAVES the attention workgroups' d bias tables there (swv2_attn_args.dbias_partials) and does not
                                touch d_bias (may be NULL): the caller sums them with swv2_cpb_bwd_multi */
    size_t dbias_part_bytes;
} swv2_block_desc;

int swv2_block_fwd(const swv2_block_desc* d, void* stream);
int swv2_block_bwd(const swv2_block_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWV2_H */
