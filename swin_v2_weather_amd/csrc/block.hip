// Host-side orchestration of one Swin block (forward: 7 launches, backward: 13) on top of the public kernels' entry
// points.  Pure C++ host code: no device code here, no allocation, no synchronisation.
#include "common.h"

namespace {

swv2_operand op(int kind, const void* p, int rows, int cols, long ld, const int32_t* rowidx = nullptr) {
    swv2_operand o = {};
    o.kind = kind; o.ptr = p; o.rowidx = rowidx; o.ld = ld; o.rows = rows; o.cols = cols;
    return o;
}
swv2_operand op_heads(const void* p, int Bw, int heads, int parts, int Lp, int DP) {
    swv2_operand o = op(SWV2_OP_HEADS, p, Bw * Lp, parts * heads * DP, parts);
    o.p[0] = heads; o.p[2] = Lp; o.p[3] = DP;
    return o;
}
swv2_epilogue epi(int kind, void* out, long ld, const float* bias = nullptr, const void* aux = nullptr,
                  float* aux_out = nullptr, const int32_t* rowidx = nullptr) {
    swv2_epilogue e = {};
    e.kind = kind; e.out = out; e.ld = ld; e.bias = bias; e.aux = aux; e.aux_out = aux_out; e.rowidx = rowidx;
    return e;
}
swv2_attn_args attn(const swv2_block_desc* d) {
    swv2_attn_args a = {};
    a.qkvh = d->qkvh; a.logit_scale = d->logit_scale; a.bias = d->bias; a.bias_pack = d->bias ? d->bias_pack : nullptr;
    a.oh = d->oh; a.lse = d->lse;
    a.Bw = d->B * d->nwh * d->nww; a.heads = d->heads; a.L = d->L; a.head_dim = d->head_dim; a.nwh = d->nwh; a.nww = d->nww;
    a.mask_thr = d->mask_thr; a.max_chunks = 64;
    return a;
}
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// ---- side stream for the weight-gradient products ------------------------------------------------------------
// The four weight-gradient GEMMs of a block depend only on (dY, X) pairs that the main chain produces one after the
// other, and nothing downstream in the block reads their results.  Each of them, like each GEMM of the main chain, runs
// at 4-8 waves per CU, so they are launched on a second HIP stream (fork after the producer, join at the end of the
// block) and share the CUs with the dX chain instead of serialising behind it.  The stream and a small ring of events
// are created once per device and live for the process (the only hidden state of the library; both are capturable).
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[8] = {};
    unsigned next = 0;
};
SideStream* side_stream() {
    static thread_local SideStream per_dev[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream& ss = per_dev[dev];
    if (!ss.s) {
        if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) != hipSuccess) { ss.s = nullptr; return nullptr; }
        for (auto& e : ss.ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return &ss;
}
// main -> side dependency: everything enqueued on `main` so far must finish before later work on the side stream
inline void fork_to(SideStream* ss, hipStream_t main) {
    hipEvent_t e = ss->ev[ss->next++ & 7];
    (void)hipEventRecord(e, main);
    (void)hipStreamWaitEvent(ss->s, e, 0);
}
inline void join_from(SideStream* ss, hipStream_t main) {
    hipEvent_t e = ss->ev[ss->next++ & 7];
    (void)hipEventRecord(e, ss->s);
    (void)hipStreamWaitEvent(main, e, 0);
}
// launch `x` as launch number `id`; bracket it with the caller's HIP events when it is the one being timed
#define LAUNCH(id, x)                                                                         \
    do {                                                                                      \
        const bool t_ = d->ev_kernel == (id) && d->ev_start && d->ev_stop;                    \
        if (t_) (void)hipEventRecord((hipEvent_t)d->ev_start, (hipStream_t)st);                     \
        TRY(x);                                                                               \
        if (t_) (void)hipEventRecord((hipEvent_t)d->ev_stop, (hipStream_t)st);                      \
    } while (0)

}  // namespace

extern "C" int swv2_block_fwd(const swv2_block_desc* d, void* st) {
    SWV2_CHECK_ARG(d && d->x && d->x2 && d->qkvh && d->rowidx, "swv2_block_fwd: null descriptor field");
    const int BT = d->B * d->T, Bw = d->B * d->nwh * d->nww, Mw = Bw * d->Lp, C = d->C, h = d->heads, hid = d->hidden;
    SWV2_CHECK_ARG(!d->fuse_attn, "swv2_block_fwd: fuse_attn is reserved (0)");
    // 1. roll + partition gather | qkv GEMM | + bias, split heads, L2-normalise q, k
    {
        swv2_operand a = op(SWV2_OP_F32, d->x, Mw, C, C, d->rowidx);
        swv2_epilogue e = epi(SWV2_EPI_QKV_HEADS, d->qkvh, 0, d->qkv_b_pad, nullptr, d->rnorm);
        e.p[0] = h; e.p[2] = d->Lp; e.p[3] = d->DP; e.p[4] = d->L;
        if (d->DP > 64) {        // wide heads (96 / 128 columns): squared norms accumulate in rnorm, finished by swv2_qk_normalize
            if (hipMemsetAsync(d->rnorm, 0, (size_t)Bw * h * 2 * d->Lp * sizeof(float), (hipStream_t)st) != hipSuccess) {
                swv2_set_error("swv2_block_fwd: hipMemsetAsync(rnorm) failed");
                return SWV2_ERR_LAUNCH;
            }
        }
        LAUNCH(1, swv2_linear(&a, d->w_qkv, &e, 3 * h * d->DP, st));
        if (d->DP > 64) TRY(swv2_qk_normalize(d->qkvh, (float*)d->rnorm, Bw, h, d->Lp, d->L, d->DP, st));
    }
    // 2. cosine attention core (the CPB table is packed once into the kernels' layouts; the backward reuses it)
    {
        if (d->bias && d->bias_pack && !d->bias_prepacked) TRY(swv2_attn_pack_bias(d->bias, h, d->L, d->bias_pack, st));
        swv2_attn_args a = attn(d);
        if (d->bias) a.max_chunks = 32;          // per-workgroup table load: fewer, longer-lived workgroups
        LAUNCH(2, swv2_attn_fwd(&a, st));
    }
    if (d->fuse_proj_ln && swv2_proj_ln_supported(C, h, d->DP)) {
        // 3-4 fused: merge heads, proj GEMM, LN1 + drop-path + residual + reverse / un-roll scatter in one kernel
        swv2_proj_ln_args m = {};
        m.oh = d->oh; m.wp = d->w_proj; m.bp = d->proj_b; m.gamma = d->n1_w; m.beta = d->n1_b; m.scale = d->dp1; m.rowidx = d->rowidx;
        m.x = d->x; m.a1 = d->a1; m.mean = d->mean1; m.rstd = d->rstd1; m.y = d->x1; m.Bw = Bw; m.Lp = d->Lp; m.heads = h; m.C = C;
        m.rows_per_sample = d->T; m.eps = 1e-5f;
        LAUNCH(3, swv2_proj_ln_fwd(&m, st));
    } else {
    // 3. merge heads | proj GEMM
    {
        swv2_operand a = op_heads(d->oh, Bw, h, 1, d->Lp, d->DP);
        swv2_epilogue e = epi(SWV2_EPI_BF16, d->a1, C, d->proj_b);
        LAUNCH(3, swv2_linear(&a, d->w_proj, &e, C, st));
    }
    // 4. LN1 + drop-path + residual, scattered through reverse + un-roll
    {
        swv2_ln_args l = {};
        l.a = d->a1; l.res = d->x; l.gamma = d->n1_w; l.beta = d->n1_b; l.scale = d->dp1; l.rowidx = d->rowidx; l.y = d->x1;
        l.mean = d->mean1; l.rstd = d->rstd1; l.M = Mw; l.C = C; l.res_mod = 0; l.rows_per_sample = d->T; l.eps = 1e-5f;
        LAUNCH(4, swv2_ln_residual_fwd(&l, st));
    }
    }
    // 5-7 fused: fc1, GELU, fc2, LN2 + drop-path + residual in one kernel (the hidden activation stays in registers)
    if (d->fuse_mlp && swv2_mlp_supported(C, hid)) {
        swv2_mlp_args m = {};
        m.x = d->x1; m.w1 = d->w_fc1; m.b1 = d->fc1_b; m.w2 = d->w_fc2; m.b2 = d->fc2_b; m.gamma = d->n2_w; m.beta = d->n2_b;
        m.scale = d->dp2; m.hpre = d->hpre; m.a2 = d->a2; m.mean = d->mean2; m.rstd = d->rstd2; m.y = d->x2;
        m.M = BT; m.C = C; m.hidden = hid; m.rows_per_sample = d->T; m.eps = 1e-5f;
        LAUNCH(5, swv2_mlp_fwd(&m, st));
        return SWV2_OK;
    }
    // 5. fc1 (+ bias -> pre-activation and GELU), 6. fc2
    {
        swv2_operand a = op(SWV2_OP_F32, d->x1, BT, C, C);
        swv2_epilogue e = epi(SWV2_EPI_BF16_GELU, d->hpre, hid, d->fc1_b, nullptr, (float*)d->hact);
        LAUNCH(5, swv2_linear(&a, d->w_fc1, &e, hid, st));
    }
    {
        swv2_operand a = op(SWV2_OP_BF16, d->hact, BT, hid, hid);
        swv2_epilogue e = epi(SWV2_EPI_BF16, d->a2, C, d->fc2_b);
        LAUNCH(6, swv2_linear(&a, d->w_fc2, &e, C, st));
    }
    // 7. LN2 + drop-path + residual
    {
        swv2_ln_args l = {};
        l.a = d->a2; l.res = d->x1; l.gamma = d->n2_w; l.beta = d->n2_b; l.scale = d->dp2; l.y = d->x2; l.mean = d->mean2;
        l.rstd = d->rstd2; l.M = BT; l.C = C; l.res_mod = 0; l.rows_per_sample = d->T; l.eps = 1e-5f;
        LAUNCH(7, swv2_ln_residual_fwd(&l, st));
    }
    return SWV2_OK;
}

extern "C" int swv2_block_bwd(const swv2_block_desc* d, void* st) {
    SWV2_CHECK_ARG(d && d->dx2 && d->dx && d->da2 && d->dh && d->dx1 && d->da1 && d->doh && d->dqkvh && d->ln_ws,
                   "swv2_block_bwd: null descriptor field");
    const int BT = d->B * d->T, Bw = d->B * d->nwh * d->nww, Mw = Bw * d->Lp, C = d->C, h = d->heads, hid = d->hidden;
    const int sp = d->wgrad_splits > 0 ? d->wgrad_splits : 64;
    SideStream* ss = d->wgrad_side_stream ? side_stream() : nullptr;
    void* ws = ss ? (void*)ss->s : st;            // stream of the weight-gradient products
    const bool fused = d->fuse_mlp && swv2_mlp_supported(C, hid);
    const bool fused_pl = d->fuse_proj_ln && swv2_proj_ln_supported(C, h, d->DP);
    // the four weight gradients as ONE launch after the data path (needs the operand set of the fused paths)
    const bool defer = fused && fused_pl && d->ln_ws_floats >= swv2_mlp_bwd_ws_floats(BT, C) + swv2_proj_ln_bwd_ws_floats(Mw, C);
    int n_ln1 = 0, n_ln2 = 0;
    // (needs the fused MLP path's operand set: GELU(hpre) on load; the proj + LN1 pair may be fused or not)
    const bool group = d->wgrad_group && fused && d->wgrad_ws && !ss &&
                       d->wgrad_ws_bytes >= swv2_block_wgrad_ws_bytes(C, hid, h * d->DP, 0);
    if (fused) {
        // 7', 6', 5' data path fused: LN2 backward, dh = (da2 W2) * GELU'(hpre), dx1 = dx2 + dh W1 in one kernel
        swv2_mlp_bwd_args m = {};
        m.dy = d->dx2; m.a2 = d->a2; m.mean = d->mean2; m.rstd = d->rstd2; m.gamma = d->n2_w; m.scale = d->dp2; m.hpre = d->hpre;
        m.w2t = d->w_fc2t; m.w1t = d->w_fc1t; m.da2 = d->da2; m.dh = d->dh; m.dx = d->dx1; m.dgamma = d->d_n2_w;
        m.dbeta = d->d_n2_b; m.ws = d->ln_ws; m.M = BT; m.C = C; m.hidden = hid; m.rows_per_sample = d->T;
        m.ws = d->ln_ws;
        LAUNCH(13, swv2_mlp_bwd_impl(&m, st, defer ? &n_ln2 : nullptr, (float*)d->grad_zero, (long)(d->grad_zero_bytes / 4)));
        // weight gradients (hact was not stored: GELU(hpre) on load)
        swv2_operand dy2 = op(SWV2_OP_BF16, d->da2, BT, C, C), x2 = op(SWV2_OP_BF16_GELU, d->hpre, BT, hid, hid);
        swv2_operand dy1 = op(SWV2_OP_BF16, d->dh, BT, hid, hid), x1 = op(SWV2_OP_F32, d->x1, BT, C, C);
        if (ss) fork_to(ss, (hipStream_t)st);
        if (!group) {
        LAUNCH(12, swv2_linear_wgrad_ws(&dy2, &x2, d->d_fc2_w, d->d_fc2_b, nullptr, nullptr, hid, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
        LAUNCH(14, swv2_linear_wgrad_ws(&dy1, &x1, d->d_fc1_w, d->d_fc1_b, nullptr, nullptr, C, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
        }
    } else {
    // 7'. LN2 backward
    {
        swv2_ln_args l = {};
        l.a = d->a2; l.dy = d->dx2; l.gamma = d->n2_w; l.scale = d->dp2; l.mean = d->mean2; l.rstd = d->rstd2; l.da = d->da2;
        l.dgamma = d->d_n2_w; l.dbeta = d->d_n2_b; l.ws = d->ln_ws; l.M = BT; l.C = C; l.rows_per_sample = d->T;
        LAUNCH(11, swv2_ln_residual_bwd(&l, st));
    }
    // 6'. fc2: dW = da2^T GELU(h) ; dh = (da2 W2) * GELU'(h)
    {
        swv2_operand dy = op(SWV2_OP_BF16, d->da2, BT, C, C),
                     x = op(SWV2_OP_BF16, d->hact, BT, hid, hid);
        if (ss) fork_to(ss, (hipStream_t)st);
        LAUNCH(12, swv2_linear_wgrad_ws(&dy, &x, d->d_fc2_w, d->d_fc2_b, nullptr, nullptr, hid, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
        swv2_epilogue e = epi(SWV2_EPI_GELU_GRAD, d->dh, hid, nullptr, d->hpre);
        LAUNCH(13, swv2_linear(&dy, d->w_fc2t, &e, hid, st));
    }
    // 5'. fc1: dW = dh^T x1 ; dx1 = dx2 + dh W1
    {
        swv2_operand dy = op(SWV2_OP_BF16, d->dh, BT, hid, hid), x = op(SWV2_OP_F32, d->x1, BT, C, C);
        if (ss) fork_to(ss, (hipStream_t)st);
        LAUNCH(14, swv2_linear_wgrad_ws(&dy, &x, d->d_fc1_w, d->d_fc1_b, nullptr, nullptr, C, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
        swv2_epilogue e = epi(SWV2_EPI_F32, d->dx1, C, nullptr, d->dx2);
        LAUNCH(15, swv2_linear(&dy, d->w_fc1t, &e, C, st));
    }
    }
    if (fused_pl) {
        // 4' + the data path of 3' fused: LN1 backward (row gather) and d(oh) = split(da1 Wp) in one kernel
        swv2_proj_ln_bwd_args m = {};
        m.dy = d->dx1; m.a1 = d->a1; m.mean = d->mean1; m.rstd = d->rstd1; m.gamma = d->n1_w; m.scale = d->dp1; m.rowidx = d->rowidx;
        m.wpt = d->w_projt; m.da1 = d->da1; m.doh = d->doh; m.dgamma = d->d_n1_w; m.dbeta = d->d_n1_b; m.ws = d->ln_ws;
        m.Bw = Bw; m.Lp = d->Lp; m.heads = h; m.C = C; m.rows_per_sample = d->T;
        if (defer) m.ws = d->ln_ws + swv2_mlp_bwd_ws_floats(BT, C);      // second region: LN2's partial rows are still unfolded
        LAUNCH(16, swv2_proj_ln_bwd_impl(&m, st, defer ? &n_ln1 : nullptr));
        swv2_operand dy = op(SWV2_OP_BF16, d->da1, Mw, C, C), x = op_heads(d->oh, Bw, h, 1, d->Lp, d->DP);
        if (ss) fork_to(ss, (hipStream_t)st);
        if (!group)
        LAUNCH(17, swv2_linear_wgrad_ws(&dy, &x, d->d_proj_w, d->d_proj_b, nullptr, d->proj_map, C, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
    } else {
    // 4'. LN1 backward (gathers dx1 rows through the window table; padded rows -> 0)
    {
        swv2_ln_args l = {};
        l.a = d->a1; l.dy = d->dx1; l.gamma = d->n1_w; l.scale = d->dp1; l.rowidx = d->rowidx; l.mean = d->mean1; l.rstd = d->rstd1;
        l.da = d->da1; l.dgamma = d->d_n1_w; l.dbeta = d->d_n1_b; l.ws = d->ln_ws; l.M = Mw; l.C = C; l.rows_per_sample = d->T;
        LAUNCH(16, swv2_ln_residual_bwd(&l, st));
    }
    // 3'. proj: dW = da1^T merge(oh) ; d(oh) = split(da1 Wp)
    {
        swv2_operand dy = op(SWV2_OP_BF16, d->da1, Mw, C, C), x = op_heads(d->oh, Bw, h, 1, d->Lp, d->DP);
        if (ss) fork_to(ss, (hipStream_t)st);
        if (!group)
        LAUNCH(17, swv2_linear_wgrad_ws(&dy, &x, d->d_proj_w, d->d_proj_b, nullptr, d->proj_map, C, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
        swv2_epilogue e = epi(SWV2_EPI_HEADS, d->doh, 0);
        e.p[0] = h; e.p[2] = d->Lp; e.p[3] = d->DP; e.p[4] = d->L;
        LAUNCH(18, swv2_linear(&dy, d->w_projt, &e, h * d->DP, st));
    }
    }
    // 2'. attention backward (incl. the backward of the q / k normalisation)
    {
        swv2_attn_args a = attn(d);
        a.doh = d->doh; a.rnorm = d->rnorm; a.dqkvh = d->dqkvh; a.dlogit_scale = d->d_logit_scale; a.dbias = d->d_bias;
        // every workgroup adds its d bias table with atomics: fewer, longer-lived workgroups (end-to-end at depth 12:
        // 16 chunks 89.9, 32 chunks 98.8, 64 chunks 97.0 samples/s)
        if (d->bias) {
            a.max_chunks = swv2_attn_bias_chunks(Bw);
            // the weight-gradient workspace is idle until the grouped launch at the end of the block: the workgroups' d bias
            // tables go there and are summed by one more launch instead of 31 K atomics per workgroup
            if (d->dbias_part) {     // the tables stay where they are written: summed for all blocks by swv2_cpb_bwd_multi
                a.dbias_ws = d->dbias_part; a.dbias_ws_bytes = d->dbias_part_bytes; a.dbias_partials = 1;
            } else if (!ss) { a.dbias_ws = d->wgrad_ws; a.dbias_ws_bytes = d->wgrad_ws_bytes; }       // (a side stream may still be using it)
        }
        // without bias at the 176-token window one workgroup (11 waves, ~90 KB of LDS) fills a CU: exactly one persistent
        // workgroup per CU (256 / heads chunks) instead of two rounds of 256 (same box: 110.4 vs 116.2 us per launch)
        if (!d->bias && d->Lp == 176 && h <= 256) a.max_chunks = 256 / h;
        LAUNCH(19, swv2_attn_bwd(&a, st));
    }
    // 1'. qkv: dW = dqkv^T gather(x) ; dx = dx1 + scatter(dqkv Wqkv)
    {
        swv2_operand dy = op_heads(d->dqkvh, Bw, h, 3, d->Lp, d->DP), x = op(SWV2_OP_F32, d->x, Mw, C, C, d->rowidx);
        if (ss) fork_to(ss, (hipStream_t)st);
        if (!group)
        LAUNCH(20, swv2_linear_wgrad_ws(&dy, &x, d->d_qkv_w, d->d_qkv_b, d->qkv_map, nullptr, C, sp, d->wgrad_ws, d->wgrad_ws_bytes, ws));
        swv2_epilogue e = epi(SWV2_EPI_F32, d->dx, C, nullptr, d->dx1, nullptr, d->rowidx);
        LAUNCH(21, swv2_linear(&dy, d->w_qkvt, &e, C, st));
    }
    // d gamma / d beta of both LayerNorms: one reduction for both -- riding on the weight-gradient reduction launch when the grouped
    // products run (swv2_block_wgrad_ln), a launch of its own otherwise
    swv2_ln_partials lnp = {};
    if (defer) {
        lnp.ws[0] = d->ln_ws; lnp.dgamma[0] = d->d_n2_w; lnp.dbeta[0] = d->d_n2_b; lnp.n[0] = n_ln2;
        lnp.ws[1] = d->ln_ws + swv2_mlp_bwd_ws_floats(BT, C); lnp.dgamma[1] = d->d_n1_w; lnp.dbeta[1] = d->d_n1_b; lnp.n[1] = n_ln1;
        lnp.C = C;
        if (!group)
            swv2_launch_ln_partials_reduce2(lnp.ws[0], lnp.dgamma[0], lnp.dbeta[0], n_ln2, lnp.ws[1], lnp.dgamma[1], lnp.dbeta[1], n_ln1, C,
                                            (hipStream_t)st);
    }
    if (group) {
        swv2_wgrad_item it[4] = {};
        it[0].dy = op(SWV2_OP_BF16, d->da2, BT, C, C); it[0].x = op(SWV2_OP_BF16_GELU, d->hpre, BT, hid, hid);
        it[0].dW = d->d_fc2_w; it[0].db = d->d_fc2_b; it[0].ldw = hid;
        it[1].dy = op(SWV2_OP_BF16, d->dh, BT, hid, hid); it[1].x = op(SWV2_OP_F32, d->x1, BT, C, C);
        it[1].dW = d->d_fc1_w; it[1].db = d->d_fc1_b; it[1].ldw = C;
        it[2].dy = op(SWV2_OP_BF16, d->da1, Mw, C, C); it[2].x = op_heads(d->oh, Bw, h, 1, d->Lp, d->DP);
        it[2].dW = d->d_proj_w; it[2].db = d->d_proj_b; it[2].kmap = d->proj_map; it[2].ldw = C;
        it[3].dy = op_heads(d->dqkvh, Bw, h, 3, d->Lp, d->DP); it[3].x = op(SWV2_OP_F32, d->x, Mw, C, C, d->rowidx);
        it[3].dW = d->d_qkv_w; it[3].db = d->d_qkv_b; it[3].nmap = d->qkv_map; it[3].ldw = C;
        LAUNCH(22, swv2_block_wgrad_ln(it, 0, d->wgrad_ws, d->wgrad_ws_bytes, defer ? &lnp : nullptr, st));
    }
    if (ss) join_from(ss, (hipStream_t)st);
    return SWV2_OK;
}
