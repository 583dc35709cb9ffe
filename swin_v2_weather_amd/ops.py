"""Thin tensor-level wrappers over the C ABI (include/swv2.h).

PyTorch is used for device memory, streams and autograd bookkeeping only; every arithmetic op of the hot path is a
HIP kernel in libswv2.so.  All wrappers enqueue on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Optional

import torch

from . import _lib as L

BF16 = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-kernel event timing (bench.py): HIP events on the launch stream around selected entry points ----
_TIMED = None          # None or dict name -> list of (start_event, end_event)


_EVERY = 1
_EVPOOL = []            # event pairs whose hipEvent_t handles exist already (materialised OUTSIDE any timed region)


def prewarm_events(n):
    """Create n HIP event pairs now (an event's handle is created by its first record, which is a packet on the stream)."""
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        _EVPOOL.append((a, b))


def _event_pair():
    if _EVPOOL:
        return _EVPOOL.pop()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); b.record()                       # materialise the hipEvent_t handles (re-recorded by the library)
    return a, b


def start_kernel_timing(names, every=1):
    """Bracket launches of the named kernels with HIP event pairs on the launch stream.  `every` = N: only every N-th eligible
    block call gets a pair -- an event record between two kernels is a packet the next kernel waits behind (measured: one pair per
    block call, 24 per step, costs 3.3 % of the benchmark step), so a timed region samples instead of bracketing everything."""
    global _TIMED, _EVERY
    _TIMED = {n: [] for n in names}
    _EVERY = max(1, int(every))
    _RR["fwd"] = _RR["bwd"] = 0


def stop_kernel_timing():
    """-> {name: (launches, mean_ms)}; synchronises."""
    global _TIMED
    rec, _TIMED = _TIMED, None
    torch.cuda.synchronize()
    out = {n: (len(ev), sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)) for n, ev in (rec or {}).items()}
    for ev in (rec or {}).values():
        _EVPOOL.extend(ev)
    return out


BLOCK_KERNEL_IDS = {"qkv": 1, "attn_fwd": 2, "proj": 3, "ln1_fwd": 4, "fc1": 5, "fc2": 6, "ln2_fwd": 7, "ln2_bwd": 11,
                    "wgrad_fc2": 12, "dh": 13, "wgrad_fc1": 14, "dx1": 15, "ln1_bwd": 16, "wgrad_proj": 17, "doh": 18,
                    "attn_bwd": 19, "wgrad_qkv": 20, "dx": 21,
                    # fused kernels share the launch slot of the first kernel they replace
                    "proj_ln_fwd": 3, "mlp_fwd": 5, "mlp_bwd": 13, "proj_ln_bwd": 16,
                    # the four weight gradients of a block as one grouped launch + its partial-tile reduction
                    "wgrad_group": 22}


_RR = {"fwd": 0, "bwd": 0}


def block_event_pair(phase, desc):
    """If kernels of this phase ("fwd": launch ids 1-7, "bwd": 11-21) are being timed (start_kernel_timing), attach a
    fresh HIP event pair to the block descriptor so that swv2_block_fwd/bwd brackets exactly ONE launch on the launch
    stream; with several requested kernels of a phase the calls take turns (round robin over the blocks of the step)."""
    desc.ev_kernel = 0
    rec = _TIMED
    if rec is None:
        return
    mine = [n for n in rec if BLOCK_KERNEL_IDS.get(n, 0) and ((BLOCK_KERNEL_IDS[n] < 10) == (phase == "fwd"))]
    if not mine:
        return
    turn = _RR[phase]
    _RR[phase] += 1
    if turn % _EVERY:
        return
    n = mine[(turn // _EVERY) % len(mine)]
    a, b = _event_pair()
    desc.ev_kernel, desc.ev_start, desc.ev_stop = BLOCK_KERNEL_IDS[n], a.cuda_event, b.cuda_event
    rec[n].append((a, b))


def _timed(name, fn, *args):
    rec = _TIMED
    if rec is None or name not in rec:
        return fn(*args)
    a, b = _event_pair()
    a.record()
    r = fn(*args)
    b.record()
    rec[name].append((a, b))
    return r


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise L.Swv2Error(f"{name}: expected a contiguous CUDA {dtype} tensor, got {t.dtype} "
                          f"contiguous={t.is_contiguous()} device={t.device}")
    return t


def attn_geometry(Lwin: int, head_dim: int):
    lp, dp = C.c_int(), C.c_int()
    L.check(L.load().swv2_attn_geometry(Lwin, head_dim, C.byref(lp), C.byref(dp)), "swv2_attn_geometry")
    return lp.value, dp.value


# ---- operand / epilogue builders ---------------------------------------------------------------------------
def operand(kind, t, rows, cols, ld=0, rowidx=None, aux=(None, None, None, None), p=(0, 0, 0, 0)) -> L.Operand:
    o = L.Operand()
    o.kind, o.ptr, o.rowidx = kind, _p(t), _p(rowidx)
    o.aux0, o.aux1, o.aux2, o.aux3 = (_p(a) for a in aux)
    o.ld, o.rows, o.cols = ld, rows, cols
    o.p = (C.c_int * 4)(*p)
    o._keep = (t, rowidx, aux)
    return o


def op_f32(x2d: torch.Tensor, rows=None, rowidx=None):
    _chk(x2d, torch.float32, "op_f32")
    return operand(L.OP_F32, x2d, rows if rows is not None else x2d.shape[0], x2d.shape[1], x2d.shape[1], rowidx)


def op_bf16(x2d: torch.Tensor, rows=None, rowidx=None, gelu=False):
    _chk(x2d, BF16, "op_bf16")
    return operand(L.OP_BF16_GELU if gelu else L.OP_BF16, x2d, rows if rows is not None else x2d.shape[0],
                   x2d.shape[1], x2d.shape[1], rowidx)


def op_bf16_cscale(x2d: torch.Tensor, coef: torch.Tensor, rows_per_sample: int, cols: int = 0):
    """bf16 rows scaled on load by coef[row // rows_per_sample][col // 16] (fp32 [B, cols / 16]); `cols` < x2d.shape[1]: only the first
    `cols` columns of every row are the operand (the loss epilogue's residual: rows padded to whole cache lines)"""
    _chk(x2d, BF16, "op_bf16_cscale"); _chk(coef, torch.float32, "op_bf16_cscale coef")
    cols = cols or x2d.shape[1]
    if coef.dim() != 2 or coef.shape[1] * 16 != cols or coef.shape[0] * rows_per_sample != x2d.shape[0] or cols > x2d.shape[1]:
        raise L.Swv2Error(f"op_bf16_cscale: coef {tuple(coef.shape)} does not match rows {x2d.shape[0]} / {rows_per_sample}, cols {cols}")
    return operand(L.OP_BF16_CSCALE, x2d, x2d.shape[0], cols, x2d.shape[1], aux=(coef, None, None, None),
                   p=(rows_per_sample, 0, coef.shape[1], 0))


def op_heads(t: torch.Tensor, Bw, heads, parts, Lp, DP):
    _chk(t, BF16, "op_heads")
    return operand(L.OP_HEADS, t, Bw * Lp, parts * heads * DP, parts, p=(heads, 0, Lp, DP))


def op_patch(x4d: torch.Tensor):
    _chk(x4d, torch.float32, "op_patch")
    B, Cin, H, W = x4d.shape
    return operand(L.OP_PATCH, x4d, B * (H // 4) * (W // 4), Cin * 16, 0, p=(Cin, H, W, 0))


def op_merge_ln(x4d, mean, rstd, gamma, beta):
    B, H, W, Cc = x4d.shape
    return operand(L.OP_MERGE_LN, x4d, B * (H // 2) * (W // 2), 4 * Cc, 0, aux=(mean, rstd, gamma, beta), p=(H, W, Cc, 0))


def epilogue(kind, out, ld=0, bias=None, aux=None, aux_out=None, rowidx=None, p=(0, 0, 0, 0, 0), loss=None) -> L.Epilogue:
    """loss (EPI_UNPATCH_LOSS) = (tar [B, Ct, H, W] fp32, quadrature row weights [H], per-group partial sums
    [ceil(M / LOSS_GROUP_ROWS), 2, Cout, 2] fp32, residual bf16 [M, N], first target channel[, base]); `out` (or, in rollouts, the
    tensor `base` that `out` is a channel slice of, with p[4] = its channels per sample), the residual and a second destination
    `aux_out` must have LOSS_DUMP_BYTES of allocated memory behind them; follow the launch with loss_part_reduce"""
    e = L.Epilogue()
    e.kind, e.out, e.bias, e.aux, e.aux_out, e.rowidx, e.ld = kind, _p(out), _p(bias), _p(aux), _p(aux_out), _p(rowidx), ld
    e.p = (C.c_int * 5)(*p)
    e._keep = (out, bias, aux, aux_out, rowidx, loss)
    if loss is not None:
        tar, qw, part, resid, coff = loss[:5]
        base = loss[5] if len(loss) > 5 else None
        _chk(tar, torch.float32, "loss target"); _chk(qw, torch.float32, "quadrature weights")
        _chk(resid, BF16, "loss residual"); _chk(part, torch.float32, "loss partial sums")
        if resid.dim() != 2 or resid.shape[1] % 64:
            raise L.Swv2Error(f"loss epilogue: residual rows must be padded to whole 128-byte lines (ops._lib.loss_resid_pitch), got {tuple(resid.shape)}")
        behind = [(out if base is None else base, "out"), (resid, "residual")] + ([(aux_out, "second destination")] if aux_out is not None else [])
        for t_, nm in behind:      # room behind the tensors for the masked lanes' stores
            room = t_.untyped_storage().nbytes() - (t_.storage_offset() + t_.numel()) * t_.element_size()
            if room < L.LOSS_DUMP_BYTES:
                raise L.Swv2Error(f"loss epilogue: {nm} needs {L.LOSS_DUMP_BYTES} bytes of allocated scratch behind it ({room} found)")
        e.loss_tar, e.loss_qw, e.loss_part, e.loss_resid = _p(tar), _p(qw), _p(part), _p(resid)
        dump = 0
        if base is not None:      # floats from `out` to the end of the tensor it is a slice of
            dump = (base.data_ptr() + base.numel() * 4 - out.data_ptr()) // 4
        e.q = (C.c_int * 3)(tar.shape[1], coff, dump)
    return e


def linear(a: L.Operand, w_bf16: torch.Tensor, e: L.Epilogue, N: int, tag: str = "linear"):
    _chk(w_bf16, BF16, "linear weight")
    if w_bf16.shape[0] != N or w_bf16.shape[1] != a.cols:
        raise L.Swv2Error(f"linear: weight {tuple(w_bf16.shape)} does not match N={N} K={a.cols}")
    L.check(_timed(tag, L.load().swv2_linear, C.byref(a), _p(w_bf16), C.byref(e), N, _stream()), "swv2_linear")


def linear_wgrad(dy: L.Operand, x: L.Operand, dW: torch.Tensor, db: Optional[torch.Tensor], nmap=None, kmap=None,
                 splits: int = 0, workspace: bool = True):
    """dW += dY^T X.  workspace=True: per-slice partial tiles + reduce kernel (deterministic, no atomics on dW), the
    partials live in a torch allocation that is released (stream-ordered) after the call; False: fp32 atomics on dW.
    splits = 0: as many row slices as fill the chip in ONE round (two 70 KB workgroups per CU = 512): the PatchEmbed / head
    gradients (10 output tiles) ran 1280 workgroups = 2.5 rounds at 128 slices and wrote 84 MB of partial tiles."""
    _chk(dW, torch.float32, "dW")
    ldw = dW.shape[-1] if dW.dim() == 2 else dW[0].numel()
    lib = L.load()
    if splits <= 0:
        tiles = ((dy.cols + 127) // 128) * ((x.cols + 127) // 128)
        splits = max(8, (512 // tiles) // 8 * 8)
        if tiles < 3:                       # the library doubles the slice count of small outputs itself
            splits = max(8, splits // 16 * 8)
    if not workspace:
        L.check(lib.swv2_linear_wgrad(C.byref(dy), C.byref(x), _p(dW), _p(db), _p(nmap), _p(kmap), ldw, splits, _stream()),
                "swv2_linear_wgrad")
        return
    nbytes = lib.swv2_linear_wgrad_ws_bytes(dy.rows, dy.cols, x.cols, splits)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dW.device)
    L.check(lib.swv2_linear_wgrad_ws(C.byref(dy), C.byref(x), _p(dW), _p(db), _p(nmap), _p(kmap), ldw, splits, _p(ws), nbytes,
                                     _stream()), "swv2_linear_wgrad_ws")


def prep_weight(w: torch.Tensor, transpose=False, row_map=None, out_rows=None, col_map=None, out_cols=None):
    """fp32 [rows][cols] parameter -> bf16 [out_rows][out_cols] (cast / transpose / permute / zero-pad)."""
    w2 = w.detach().reshape(w.shape[0], -1)
    _chk(w2, torch.float32, "prep_weight")
    rows, cols = w2.shape
    r_t, c_t = (cols, rows) if transpose else (rows, cols)
    out_rows = out_rows if out_rows is not None else r_t
    out_cols = out_cols if out_cols is not None else c_t
    out = torch.empty(out_rows, out_cols, dtype=BF16, device=w.device)
    L.check(L.load().swv2_prep_weight(_p(w2), rows, cols, int(transpose), _p(row_map), out_rows, _p(col_map), out_cols,
                                      _p(out), _stream()), "swv2_prep_weight")
    return out


class PrepBatch:
    """Collects parameter preparations (prep_weight's arguments) and runs them as ONE swv2_prep_multi launch.  `add` returns
    the output tensor at once (a given `out` of the right shape / dtype is reused, so the pointers handed to the kernels
    stay stable); `launch` must be called before anything reads the outputs.  The device tables are cached per job-list
    identity: in steady state (same parameters, same outputs every step) nothing is uploaded."""

    _tables = {}

    def __init__(self):
        self.jobs, self.keep, self.done = [], [], []

    def on_launched(self, fn):
        """fn() runs after the launch that covers the jobs added so far has been enqueued successfully"""
        self.done.append(fn)

    def add(self, w: torch.Tensor, transpose=False, row_map=None, out_rows=None, col_map=None, out_cols=None, out=None, f32=False):
        w2 = w.detach().reshape(w.shape[0], -1)
        _chk(w2, torch.float32, "PrepBatch.add")
        rows, cols = w2.shape
        r_t, c_t = (cols, rows) if transpose else (rows, cols)
        out_rows = out_rows if out_rows is not None else r_t
        out_cols = out_cols if out_cols is not None else c_t
        dt = torch.float32 if f32 else BF16
        if out is None or out.dtype != dt or tuple(out.shape) != (out_rows, out_cols) or out.device != w.device or not out.is_contiguous():
            out = torch.empty(out_rows, out_cols, dtype=dt, device=w.device)
        self.jobs.append((_p(w2), rows, cols, int(transpose), _p(row_map), out_rows, _p(col_map), out_cols, _p(out), int(f32)))
        self.keep.append((w2, row_map, col_map, out))
        return out

    def launch(self):
        if not self.jobs:
            for fn in self.done:
                fn()
            self.done = []
            return
        lib = L.load()
        key = tuple(self.jobs)
        t = PrepBatch._tables.get(key)
        dev = self.keep[0][3].device
        if t is None:
            arr = (L.PrepItem * len(self.jobs))()
            pairs = []
            for i, j in enumerate(self.jobs):
                arr[i] = L.PrepItem(*j)                        # (w, rows, cols, transpose, row_map, out_rows, col_map, out_cols, out, f32)
                pairs += [(i, c) for c in range(lib.swv2_prep_item_chunks(j[5], j[7], j[3]))]
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            t = (host.to(dev), torch.tensor(pairs, dtype=torch.int32).to(dev), len(pairs))
            if len(PrepBatch._tables) > 64:
                PrepBatch._tables.clear()
            PrepBatch._tables[key] = t
        L.check(lib.swv2_prep_multi(_p(t[0]), _p(t[1]), t[2], _stream()), "swv2_prep_multi")
        for fn in self.done:
            fn()
        self.jobs, self.keep, self.done = [], [], []


def ln_residual_fwd(a, res, gamma, beta, scale, rowidx, y, mean, rstd, M, Cc, res_mod, rows_per_sample, eps=1e-5):
    g = L.LnArgs()
    g.a, g.res, g.gamma, g.beta, g.scale, g.rowidx = _p(a), _p(res), _p(gamma), _p(beta), _p(scale), _p(rowidx)
    g.y, g.mean, g.rstd = _p(y), _p(mean), _p(rstd)
    g.M, g.C, g.res_mod, g.rows_per_sample, g.eps = M, Cc, res_mod, rows_per_sample, eps
    L.check(L.load().swv2_ln_residual_fwd(C.byref(g), _stream()), "swv2_ln_residual_fwd")


def ln_residual_bwd(a, dy, gamma, scale, rowidx, mean, rstd, da, dgamma, dbeta, M, Cc, rows_per_sample):
    g = L.LnArgs()
    g.a, g.dy, g.gamma, g.scale, g.rowidx = _p(a), _p(dy), _p(gamma), _p(scale), _p(rowidx)
    g.mean, g.rstd, g.da, g.dgamma, g.dbeta = _p(mean), _p(rstd), _p(da), _p(dgamma), _p(dbeta)
    ws = torch.empty(L.LN_BWD_MAX_BLOCKS * 2 * Cc, dtype=torch.float32, device=a.device)
    g.ws = _p(ws)
    g.M, g.C, g.res_mod, g.rows_per_sample, g.eps = M, Cc, 0, rows_per_sample, 1e-5
    L.check(L.load().swv2_ln_residual_bwd(C.byref(g), _stream()), "swv2_ln_residual_bwd")


def attn_args(qkvh, logit_scale, bias, oh, lse, Bw, heads, Lwin, head_dim, nwh, nww, mask_thr, doh=None, rnorm=None,
              dqkvh=None, dlogit=None, dbias=None, max_chunks=64, bias_pack=None, dbias_ws=None) -> L.AttnArgs:
    a = L.AttnArgs()
    a.qkvh, a.logit_scale, a.bias, a.oh, a.lse = _p(qkvh), _p(logit_scale), _p(bias), _p(oh), _p(lse)
    a.bias_pack = _p(bias_pack)
    a.doh, a.rnorm, a.dqkvh, a.dlogit_scale, a.dbias = _p(doh), _p(rnorm), _p(dqkvh), _p(dlogit), _p(dbias)
    a.Bw, a.heads, a.L, a.head_dim, a.nwh, a.nww, a.mask_thr, a.max_chunks = Bw, heads, Lwin, head_dim, nwh, nww, mask_thr, max_chunks
    if dbias_ws is not None:        # scratch for the workgroups' d bias tables (swv2_attn_dbias_ws_bytes); the caller keeps it alive
        a.dbias_ws, a.dbias_ws_bytes = _p(dbias_ws), dbias_ws.numel() * dbias_ws.element_size()
    return a


def attn_pack_bias(bias: torch.Tensor) -> torch.Tensor:
    """[heads][L][L] fp32 CPB table -> the attention kernels' pre-packed layouts (one uint8 buffer)."""
    heads, Lw = bias.shape[0], bias.shape[1]
    out = torch.empty(L.load().swv2_attn_pack_bias_bytes(heads, Lw), dtype=torch.uint8, device=bias.device)
    L.check(L.load().swv2_attn_pack_bias(_p(bias), heads, Lw, _p(out), _stream()), "swv2_attn_pack_bias")
    return out


def attn_fwd(a: L.AttnArgs):
    L.check(_timed("attn_fwd", L.load().swv2_attn_fwd, C.byref(a), _stream()), "swv2_attn_fwd")


def attn_bwd(a: L.AttnArgs):
    L.check(_timed("attn_bwd", L.load().swv2_attn_bwd, C.byref(a), _stream()), "swv2_attn_bwd")


def batch_sum(inp: torch.Tensor, out: torch.Tensor, accumulate=False):
    B = inp.shape[0]
    n = inp[0].numel()
    L.check(L.load().swv2_batch_sum(_p(inp), _p(out), B, n, int(accumulate), _stream()), "swv2_batch_sum")


def merge_stats(x, mean, rstd, eps=1e-5):
    B, H, W, Cc = x.shape
    L.check(L.load().swv2_merge_stats(_p(x), _p(mean), _p(rstd), B, H, W, Cc, eps, _stream()), "swv2_merge_stats")


def merge_ln_bwd(x, dn, gamma, mean, rstd, dx, dgamma, dbeta):
    B, H, W, Cc = x.shape
    L.check(L.load().swv2_merge_ln_bwd(_p(x), _p(dn), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
                                       B, H, W, Cc, _stream()), "swv2_merge_ln_bwd")


def loss_sums(prd, tar, qw, sums):
    B, Cc, H, W = prd.shape
    L.check(L.load().swv2_loss_sums(_p(prd), _p(tar), _p(qw), _p(sums), B * Cc, H, W, _stream()), "swv2_loss_sums")


def loss_finalize(sums, chw, absolute: bool, squared: bool, loss, coef):
    """sums [B, C, 2] or [layers, B, C, 2] (layers added in order)"""
    layers = sums.shape[0] if sums.dim() == 4 else 1
    BC, Cc = sums.shape[-3] * sums.shape[-2], chw.numel()
    L.check(L.load().swv2_loss_finalize(_p(sums), layers, _p(chw), BC, Cc, int(absolute), int(squared), _p(loss), _p(coef), _stream()),
            "swv2_loss_finalize")


def loss_part_reduce(part, M, T, B, Cout, coff, sums):
    """per-group partial sums of the head's loss epilogue -> sums [LOSS_PART_SLICES, B, Ct, 2] (channels coff .. coff + Cout)"""
    L.check(L.load().swv2_loss_part_reduce(_p(part), M, T, B, Cout, sums.shape[2], coff, _p(sums), _stream()), "swv2_loss_part_reduce")


def loss_resid_to_image(resid, coef, out, Cout: int, add=None):
    """out[:, :Cout] = coef[b, c] * un-patchify(resid) (+ add[:, :Cout]); resid bf16 [B * T, Cout * 16], out fp32 [B, Cs, H, W] contiguous,
    add fp32 [B, Cadd, H, W] contiguous or None"""
    _chk(resid, BF16, "loss residual"); _chk(coef, torch.float32, "coef"); _chk(out, torch.float32, "out")
    B, Cs, H, W = out.shape
    if resid.shape[1] != L.loss_resid_pitch(Cout * 16):
        raise L.Swv2Error(f"loss_resid_to_image: residual rows of {resid.shape[1]} elements, expected the padded pitch {L.loss_resid_pitch(Cout * 16)}")
    if add is not None:
        _chk(add, torch.float32, "add")
    L.check(L.load().swv2_loss_resid_to_image(_p(resid), _p(coef), _p(add), _p(out), B, Cout, H, W, Cs, add.shape[1] if add is not None else 0,
                                              _stream()), "swv2_loss_resid_to_image")


def loss_finalize_sums(sums, chw, absolute: bool, squared: bool):
    """[B, C, 2] (or [layers, B, C, 2]) quadrature sums -> (loss [1], coef [B, C] = 2 d loss / d S0) in one launch"""
    B, Cc = sums.shape[-3:-1]
    buf = torch.empty(B * Cc + 1, dtype=torch.float32, device=sums.device)
    coef, loss = buf[:B * Cc].view(B, Cc), buf[B * Cc:]
    loss_finalize(sums, chw, absolute, squared, loss, coef)
    return loss, coef


def loss_grad(prd, tar, qw, coef, dprd):
    B, Cc, H, W = prd.shape
    L.check(L.load().swv2_loss_grad(_p(prd), _p(tar), _p(qw), _p(coef), _p(dprd), B * Cc, H, W, _stream()), "swv2_loss_grad")


def era5_select_normalize(raw, out, chan, mean, std, coff=0, stream=None):
    """raw [B, S, Craw, Hraw, Wraw] fp32 -> out[:, coff : coff + S*len(chan)] = (raw[:, s, chan] - mean) / std, cropped to out's H, W"""
    B, S, Craw, Hraw, Wraw = raw.shape
    _, Ct, H, W = out.shape
    _chk(raw, torch.float32, "era5 raw"); _chk(out, torch.float32, "era5 out")
    L.check(L.load().swv2_era5_select_normalize(_p(raw), _p(out), _p(chan), _p(mean), _p(std), B, S, chan.numel(), Craw, Hraw, Wraw,
                                                H, W, Ct, coff, stream if stream is not None else _stream()), "swv2_era5_select_normalize")


def era5_zenith(out, sun, coff, stream=None):
    """sun [B, nz, 3] fp32 = (sin dec, cos dec, hour angle at longitude 0) per time point (utils/data_loader_era5.sun_position)
    -> out[:, coff : coff + nz] = cos of the solar zenith angle"""
    B, Ct, H, W = out.shape
    _chk(sun, torch.float32, "era5_zenith sun")
    if sun.dim() != 3 or sun.shape[0] != B or sun.shape[2] != 3:
        raise L.Swv2Error(f"era5_zenith: sun must be [B, nz, 3], got {tuple(sun.shape)}")
    L.check(L.load().swv2_era5_zenith(_p(out), _p(sun), B, sun.shape[1], H, W, Ct, coff,
                                      stream if stream is not None else _stream()), "swv2_era5_zenith")


def era5_static(stat, out, coff, stream=None):
    """stat [Cs, H, W] fp32 -> out[:, coff : coff + Cs] (broadcast over the batch)"""
    B, Ct, H, W = out.shape
    L.check(L.load().swv2_era5_static(_p(stat), _p(out), B, stat.shape[0], H, W, Ct, coff,
                                      stream if stream is not None else _stream()), "swv2_era5_static")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_inv_scale=1.0):
    L.check(L.load().swv2_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_inv_scale,
                                    _stream()), "swv2_adam_step")


def mlp_fwd(x, w1, b1, w2, b2, gamma, beta, scale, rows_per_sample, eps=1e-5, keep_hpre=True):
    """Fused fc1 -> GELU -> fc2 -> LayerNorm -> drop-path -> +x.  Returns (y, hpre, a2, mean, rstd); keep_hpre=False: the
    pre-activation is not written (hpre = None; the backward's recompute mode rebuilds it from x)."""
    M, Cc = x.shape
    hid = w1.shape[0]
    _chk(x, torch.float32, "mlp x"); _chk(w1, BF16, "mlp w1"); _chk(w2, BF16, "mlp w2")
    dev = x.device
    y = torch.empty(M, Cc, dtype=torch.float32, device=dev)
    hpre = torch.empty(M, hid, dtype=BF16, device=dev) if keep_hpre else None
    a2 = torch.empty(M, Cc, dtype=BF16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    a = L.MlpArgs()
    a.x, a.w1, a.b1, a.w2, a.b2, a.gamma, a.beta, a.scale = (_p(t) for t in (x, w1, b1, w2, b2, gamma, beta, scale))
    a.hpre, a.a2, a.mean, a.rstd, a.y = (_p(t) for t in (hpre, a2, mean, rstd, y))
    a.M, a.C, a.hidden, a.rows_per_sample, a.eps = M, Cc, hid, rows_per_sample, eps
    L.check(_timed("mlp_fwd", L.load().swv2_mlp_fwd, C.byref(a), _stream()), "swv2_mlp_fwd")
    return y, hpre, a2, mean, rstd


def mlp_bwd(dy, a2, mean, rstd, gamma, scale, hpre, w2t, w1t, dgamma, dbeta, rows_per_sample, x=None, w1=None, b1=None):
    """Fused LN backward -> dh = (da2 W2) * GELU'(hpre) -> dx = dy + dh W1.  Returns (dx, da2, dh); dgamma/dbeta accumulated.
    hpre=None: recompute mode -- the pre-activation is rebuilt from the forward's input x, fc1.weight w1 (bf16 [hid, C]), b1."""
    M, Cc = dy.shape
    hid = w2t.shape[0]
    dev = dy.device
    da2 = torch.empty(M, Cc, dtype=BF16, device=dev)
    dh = torch.empty(M, hid, dtype=BF16, device=dev)
    dx = torch.empty(M, Cc, dtype=torch.float32, device=dev)
    ws = torch.empty(L.load().swv2_mlp_bwd_ws_floats(M, Cc), dtype=torch.float32, device=dev)
    a = L.MlpBwdArgs()
    a.dy, a.a2, a.mean, a.rstd, a.gamma, a.scale, a.hpre, a.w2t, a.w1t = (_p(t) for t in (dy, a2, mean, rstd, gamma, scale, hpre, w2t, w1t))
    a.da2, a.dh, a.dx, a.dgamma, a.dbeta, a.ws = (_p(t) for t in (da2, dh, dx, dgamma, dbeta, ws))
    a.M, a.C, a.hidden, a.rows_per_sample = M, Cc, hid, rows_per_sample
    a.x, a.w1, a.b1 = _p(x), _p(w1), _p(b1)
    L.check(_timed("mlp_bwd", L.load().swv2_mlp_bwd, C.byref(a), _stream()), "swv2_mlp_bwd")
    return dx, da2, dh


def cpb_fwd(w1, b1, w2, b2, keep, bias, wh, ww, heads, hidden, drop_p):
    L.check(L.load().swv2_cpb_fwd(_p(w1), _p(b1), _p(w2), _p(b2), _p(keep), _p(bias), wh, ww, heads, hidden, drop_p, _stream()),
            "swv2_cpb_fwd")


_SCRATCH = {}


def _scratch(nbytes: int, device, tag: str) -> torch.Tensor:
    """a cached uint8 workspace per (device, tag), grown on demand: kernels that only need scratch for the duration of their own
    launches on the current stream share it instead of a torch.empty per call in the training step"""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag, _stream())
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        t = _SCRATCH[key] = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
    return t


def cpb_bwd(dbias, w1, b1, w2, keep, dw1, db1, dw2, db2, wh, ww, heads, hidden, drop_p, atomics=False):
    """d(meta MLP) accumulated into dw1 / db1 / dw2 / db2: partial rows in a scratch buffer + a fixed-order fold (default), or the float
    atomics of swv2_cpb_bwd (`atomics=True`)"""
    lib = L.load()
    if atomics:
        L.check(lib.swv2_cpb_bwd(_p(dbias), _p(w1), _p(b1), _p(w2), _p(keep), _p(dw1), _p(db1), _p(dw2), _p(db2), wh, ww,
                                 heads, hidden, drop_p, _stream()), "swv2_cpb_bwd")
        return
    nb = lib.swv2_cpb_bwd_ws_bytes(wh, ww, heads, hidden)
    ws = _scratch(nb, dbias.device, "cpb_bwd")
    L.check(lib.swv2_cpb_bwd_ws(_p(dbias), _p(w1), _p(b1), _p(w2), _p(keep), _p(dw1), _p(db1), _p(dw2), _p(db2), wh, ww,
                                heads, hidden, drop_p, _p(ws), nb, _stream()), "swv2_cpb_bwd_ws")


def cpb_multi_supported(heads: int, hidden: int, drop_p: float) -> bool:
    """shapes the one-launch-per-stage CPB kernels cover (swv2_cpb_fwd_multi / _bwd_multi: MFMA tiles of 16 heads x 64-unit waves; the
    keep words express Dropout(0.125))"""
    return 0 < heads <= 16 and hidden in (64, 128, 256, 384, 512) and abs(drop_p - 0.125) < 1e-6


def cpb_fwd_multi(ptab, nblk, keep_bits, bias_all, wh, ww, heads, hidden, drop_p):
    """bias_all [nblk][heads][L][L] from the blocks' meta MLPs (device pointer table ptab [nblk][4]); keep_bits int32 [nblk][L^2][hidden / 8]
    of random bits or None (eval)"""
    _chk(bias_all, torch.float32, "cpb_fwd_multi bias")
    if keep_bits is not None:
        _chk(keep_bits, torch.int32, "cpb_fwd_multi keep_bits")
    L.check(L.load().swv2_cpb_fwd_multi(_p(ptab), nblk, _p(keep_bits), _p(bias_all), wh, ww, heads, hidden, drop_p, _stream()),
            "swv2_cpb_fwd_multi")


def cpb_bwd_multi(dtables, nchunk, ptab, nblk, keep_bits, grads, wh, ww, heads, hidden, drop_p):
    """grads [nblk][3 hidden + heads hidden + heads] += d(meta MLP) of every block from dtables [nblk][nchunk][heads][L][L] (summed over nchunk)"""
    lib = L.load()
    _chk(dtables, torch.float32, "cpb_bwd_multi tables"); _chk(grads, torch.float32, "cpb_bwd_multi grads")
    nb = lib.swv2_cpb_bwd_multi_ws_bytes(nblk, wh, ww, heads, hidden)
    ws = _scratch(nb, grads.device, "cpb_bwd_multi")
    L.check(lib.swv2_cpb_bwd_multi(_p(dtables), nchunk, _p(ptab), nblk, _p(keep_bits), _p(grads), wh, ww, heads, hidden, drop_p,
                                   _p(ws), nb, _stream()), "swv2_cpb_bwd_multi")


def attn_pack_bias_multi(bias_all: torch.Tensor) -> torch.Tensor:
    """[ntab][heads][L][L] fp32 CPB tables -> [ntab][swv2_attn_pack_bias_bytes] uint8, one launch"""
    _chk(bias_all, torch.float32, "attn_pack_bias_multi")
    ntab, heads, Lw = bias_all.shape[0], bias_all.shape[1], bias_all.shape[2]
    nb = L.load().swv2_attn_pack_bias_bytes(heads, Lw)
    out = torch.empty(ntab, nb, dtype=torch.uint8, device=bias_all.device)
    L.check(L.load().swv2_attn_pack_bias_multi(_p(bias_all), ntab, heads, Lw, _p(out), _stream()), "swv2_attn_pack_bias_multi")
    return out


# ---- per-geometry index tables ------------------------------------------------------------------------------
class WindowPlan:
    """Index tables for one (batch, grid, window, shift, heads) geometry: the cyclic roll, the window partition, its
    inverse and the attention mask are all reduced to one int32 row table + one threshold."""

    def __init__(self, B, gh, gw, wh, ww, sh, sw, heads, head_dim, device):
        self.B, self.gh, self.gw, self.wh, self.ww, self.sh, self.sw = B, gh, gw, wh, ww, sh, sw
        self.heads, self.d = heads, head_dim
        self.L = wh * ww
        self.Lp, self.DP = attn_geometry(self.L, head_dim)
        self.nwh, self.nww = gh // wh, gw // ww
        self.nW = self.nwh * self.nww
        self.Bw = B * self.nW
        self.T = gh * gw
        if gh % wh or gw % ww:
            raise L.Swv2Error(f"window {wh}x{ww} does not divide the patch grid {gh}x{gw}")
        # swinv2_global.py:403-424 closed form: only a row shift produces a (non-zero) mask
        self.mask_thr = (wh - sh) * ww if sh > 0 else 0
        wi = torch.arange(self.nwh).view(-1, 1, 1, 1)
        wj = torch.arange(self.nww).view(1, -1, 1, 1)
        r = torch.arange(wh).view(1, 1, -1, 1)
        c = torch.arange(ww).view(1, 1, 1, -1)
        src = (((wi * wh + r + sh) % gh) * gw + (wj * ww + c + sw) % gw).reshape(self.nW, self.L)
        tab = torch.full((B, self.nW, self.Lp), -1, dtype=torch.int64)
        tab[:, :, :self.L] = src.unsqueeze(0) + (torch.arange(B) * self.T).view(B, 1, 1)
        self.rowidx = tab.reshape(-1).to(torch.int32).to(device)
        # head padding maps: padded feature (part*h + head)*DP + j -> part*C + head*d + j (or -1)
        Cc = heads * head_dim
        j = torch.arange(self.DP)
        hd = torch.arange(heads).view(-1, 1)
        one = torch.where(j.view(1, -1) < head_dim, hd * head_dim + j.view(1, -1), torch.full((1, 1), -1)).reshape(-1)
        self.proj_map = one.to(torch.int32).to(device)                          # [h*DP]
        three = torch.cat([torch.where(one >= 0, one + part * Cc, one) for part in range(3)])
        self.qkv_map = three.to(torch.int32).to(device)                         # [3*h*DP]


@functools.lru_cache(maxsize=64)
def window_plan(B, gh, gw, wh, ww, sh, sw, heads, head_dim, device_index) -> WindowPlan:
    return WindowPlan(B, gh, gw, wh, ww, sh, sw, heads, head_dim, torch.device("cuda", device_index))
