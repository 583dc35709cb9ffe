"""ctypes binding of libswv2.so (the C ABI declared in include/swv2.h).

There is NO fallback: if the shared library is missing or a call fails this module raises.  The library is built
in-tree by `build_library()` (hipcc, --offload-arch=gfx950), which `__graft_entry__.build()` calls.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("SWV2_LIB") or os.path.join(HERE, "libswv2.so")     # SWV2_LIB: a privately built variant (tools/ab_macro.sh)
SOURCES = ["capi.hip", "attn.hip", "attn2.hip", "attn_bwd_stream.hip", "attn_wide.hip", "gemm.hip", "gemm_tn.hip", "gemm_tn_slab.hip", "rowops.hip", "block.hip", "cpb.hip", "mlp.hip", "proj_ln.hip", "dataio.hip"]

ABI_VERSION = 107          # SWV2_VERSION of the include/swv2.h these ctypes mirrors were written against (checked in load())

_lib = None
_lock = threading.Lock()


class Swv2Error(RuntimeError):
    pass


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into swin_v2_weather_amd/libswv2.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "attn_common.h"), os.path.join(HERE, "..", "include", "swv2.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise Swv2Error("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise Swv2Error("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    return LIB_PATH


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources + the C header: profiles/rNN_pmc_hbm.json records it, bench.py reports
    counter traffic only when it matches the library it is timing (same function as profiles/summarize.py::source_hash)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*"))) + [os.path.join(HERE, "..", "include", "swv2.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# ---- structures (mirror include/swv2.h) ---------------------------------------------------------------------
class AttnArgs(C.Structure):
    _fields_ = [("qkvh", C.c_void_p), ("logit_scale", C.c_void_p), ("bias", C.c_void_p), ("bias_pack", C.c_void_p), ("oh", C.c_void_p),
                ("lse", C.c_void_p), ("doh", C.c_void_p), ("rnorm", C.c_void_p), ("dqkvh", C.c_void_p),
                ("dlogit_scale", C.c_void_p), ("dbias", C.c_void_p),
                ("Bw", C.c_int), ("heads", C.c_int), ("L", C.c_int), ("head_dim", C.c_int),
                ("nwh", C.c_int), ("nww", C.c_int), ("mask_thr", C.c_int), ("max_chunks", C.c_int), ("dbg", C.c_int),
                ("dbias_ws", C.c_void_p), ("dbias_ws_bytes", C.c_size_t), ("dbias_partials", C.c_int)]


class Operand(C.Structure):
    _fields_ = [("kind", C.c_int), ("ptr", C.c_void_p), ("rowidx", C.c_void_p), ("aux0", C.c_void_p),
                ("aux1", C.c_void_p), ("aux2", C.c_void_p), ("aux3", C.c_void_p), ("ld", C.c_long),
                ("rows", C.c_int), ("cols", C.c_int), ("p", C.c_int * 4)]


class PrepItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("transpose", C.c_int), ("row_map", C.c_void_p),
                ("out_rows", C.c_int), ("col_map", C.c_void_p), ("out_cols", C.c_int), ("out", C.c_void_p), ("out_f32", C.c_int)]


class WgradItem(C.Structure):
    _fields_ = [("dy", Operand), ("x", Operand), ("dW", C.c_void_p), ("db", C.c_void_p), ("nmap", C.c_void_p),
                ("kmap", C.c_void_p), ("ldw", C.c_int)]


class Epilogue(C.Structure):
    _fields_ = [("kind", C.c_int), ("out", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p),
                ("aux_out", C.c_void_p), ("rowidx", C.c_void_p), ("ld", C.c_long), ("p", C.c_int * 5),
                ("loss_tar", C.c_void_p), ("loss_qw", C.c_void_p), ("loss_part", C.c_void_p),
                ("loss_resid", C.c_void_p),
                ("q", C.c_int * 3)]


class ProjLnArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("oh", "wp", "bp", "gamma", "beta", "scale", "rowidx", "x", "a1", "mean", "rstd", "y")] + \
               [(n, C.c_int) for n in ("Bw", "Lp", "heads", "C", "rows_per_sample")] + [("eps", C.c_float)]


class ProjLnBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dy", "a1", "mean", "rstd", "gamma", "scale", "rowidx", "wpt", "da1", "doh", "dgamma", "dbeta",
                                           "ws")] + [(n, C.c_int) for n in ("Bw", "Lp", "heads", "C", "rows_per_sample")]


class MlpArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "w1", "b1", "w2", "b2", "gamma", "beta", "scale", "hpre", "a2", "mean", "rstd", "y")] + \
               [(n, C.c_int) for n in ("M", "C", "hidden", "rows_per_sample")] + [("eps", C.c_float)]


class MlpBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dy", "a2", "mean", "rstd", "gamma", "scale", "hpre", "w2t", "w1t", "da2", "dh", "dx",
                                           "dgamma", "dbeta", "ws")] + [(n, C.c_int) for n in ("M", "C", "hidden", "rows_per_sample")] + \
               [(n, C.c_void_p) for n in ("x", "w1", "b1")]


class LnArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("res", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("scale", C.c_void_p), ("rowidx", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("dy", C.c_void_p), ("da", C.c_void_p), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p), ("ws", C.c_void_p), ("M", C.c_int), ("C", C.c_int), ("res_mod", C.c_int),
                ("rows_per_sample", C.c_int), ("eps", C.c_float)]


LN_BWD_MAX_BLOCKS = 512
ATTN_FIRST_GEN, ATTN_PLAIN_STATS, ATTN_BWD_TWO_PHASE = 16, 8192, 32      # swv2_attn_args.dbg switches (SWV2_ATTN_FIRST_GEN / _PLAIN_STATS)
LOSS_PART_SLICES = 8          # SWV2_LOSS_PART_SLICES
LOSS_GROUP_ROWS = 32          # SWV2_LOSS_GROUP_ROWS
LOSS_DUMP_BYTES = 2048        # SWV2_LOSS_DUMP_BYTES


def loss_resid_pitch(n: int) -> int:
    """SWV2_LOSS_RESID_PITCH: row pitch (elements) of the loss epilogue's residual for n columns"""
    return (n + 63) // 64 * 64
class BlockDesc(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("B", "T", "C", "heads", "head_dim", "hidden", "L", "Lp", "DP", "nwh", "nww", "mask_thr")] +
                [(n, C.c_void_p) for n in (
                    "rowidx", "qkv_map", "proj_map",
                    "logit_scale", "qkv_b_pad", "proj_b", "n1_w", "n1_b", "fc1_b", "fc2_b", "n2_w", "n2_b",
                    "w_qkv", "w_proj", "w_fc1", "w_fc2", "w_qkvt", "w_projt", "w_fc1t", "w_fc2t",
                    "x", "bias", "bias_pack", "dp1", "dp2",
                    "qkvh", "rnorm", "oh", "lse", "a1", "mean1", "rstd1", "x1", "hpre", "hact", "a2", "mean2", "rstd2", "x2",
                    "dx2", "da2", "dh", "da1", "doh", "dqkvh", "dx1", "ln_ws", "dx",
                    "d_logit_scale", "d_bias", "d_qkv_w", "d_qkv_b", "d_proj_w", "d_proj_b", "d_n1_w", "d_n1_b", "d_fc1_w",
                    "d_fc1_b", "d_fc2_w", "d_fc2_b", "d_n2_w", "d_n2_b")] +
                [("wgrad_splits", C.c_int), ("ev_kernel", C.c_int), ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p),
                 ("fuse_proj_ln", C.c_int), ("fuse_attn", C.c_int), ("fuse_mlp", C.c_int), ("wgrad_ws", C.c_void_p), ("wgrad_ws_bytes", C.c_size_t), ("wgrad_side_stream", C.c_int), ("wgrad_group", C.c_int), ("grad_zero", C.c_void_p),
                 ("grad_zero_bytes", C.c_size_t), ("ln_ws_floats", C.c_size_t),
                 ("bias_prepacked", C.c_int), ("dbias_part", C.c_void_p), ("dbias_part_bytes", C.c_size_t)])


OP_F32, OP_BF16, OP_BF16_GELU, OP_HEADS, OP_PATCH, OP_MERGE_LN, OP_BF16_CSCALE = range(7)
EPI_BF16, EPI_F32, EPI_QKV_HEADS, EPI_GELU_GRAD, EPI_UNPATCH, EPI_HEADS, EPI_F32_ACC, EPI_BF16_GELU, EPI_UNPATCH_LOSS = range(9)

# every symbol include/swv2.h declares: (name, restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
SYMBOLS = {
    "swv2_version": (_I, []),
    "swv2_last_error": (C.c_char_p, []),
    "swv2_attn_geometry": (_I, [_I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "swv2_attn_fwd_regime": (_I, [_I, _I, _I, _I]),
    "swv2_attn_pack_bias_bytes": (C.c_size_t, [_I, _I]),
    "swv2_attn_dbias_ws_bytes": (C.c_size_t, [_I, _I, _I]),
    "swv2_attn_pack_bias": (_I, [_P, _I, _I, _P, _P]),
    "swv2_attn_pack_bias_multi": (_I, [_P, _I, _I, _I, _P, _P]),
    "swv2_attn_bias_chunks": (_I, [_I]),
    "swv2_attn_fwd": (_I, [C.POINTER(AttnArgs), _P]),
    "swv2_attn_bwd": (_I, [C.POINTER(AttnArgs), _P]),
    "swv2_linear": (_I, [C.POINTER(Operand), _P, C.POINTER(Epilogue), _I, _P]),
    "swv2_linear_wgrad": (_I, [C.POINTER(Operand), C.POINTER(Operand), _P, _P, _P, _P, _I, _I, _P]),
    "swv2_linear_wgrad_ws_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "swv2_qk_normalize": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "swv2_block_wgrad_ws_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "swv2_block_wgrad": (_I, [C.POINTER(WgradItem), _I, _P, C.c_size_t, _P]),
    "swv2_linear_wgrad_ws": (_I, [C.POINTER(Operand), C.POINTER(Operand), _P, _P, _P, _P, _I, _I, _P, C.c_size_t, _P]),
    "swv2_prep_weight": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P, _P]),
    "swv2_ln_residual_fwd": (_I, [C.POINTER(LnArgs), _P]),
    "swv2_ln_residual_bwd": (_I, [C.POINTER(LnArgs), _P]),
    "swv2_batch_sum": (_I, [_P, _P, _I, _L, _I, _P]),
    "swv2_merge_stats": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "swv2_merge_ln_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "swv2_loss_sums": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "swv2_loss_grad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "swv2_loss_part_reduce": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "swv2_loss_resid_to_image": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "swv2_loss_finalize": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P]),
    "swv2_prep_chunk": (_I, []),
    "swv2_prep_item_chunks": (_I, [_I, _I, _I]),
    "swv2_prep_multi": (_I, [_P, _P, _I, _P]),
    "swv2_adam_chunk": (_I, []),
    "swv2_adam_multi": (_I, [_P, _P, _I, _F, _F, _F, _F, _I, _F, _P]),
    "swv2_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P]),
    "swv2_era5_select_normalize": (_I, [_P, _P, _P, _P, _P] + [_I] * 10 + [_P]),
    "swv2_era5_zenith": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "swv2_era5_static": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "swv2_proj_ln_supported": (_I, [_I, _I, _I]),
    "swv2_proj_ln_bwd_ws_floats": (C.c_size_t, [_I, _I]),
    "swv2_proj_ln_fwd": (_I, [C.POINTER(ProjLnArgs), _P]),
    "swv2_proj_ln_bwd": (_I, [C.POINTER(ProjLnBwdArgs), _P]),
    "swv2_mlp_supported": (_I, [_I, _I]),
    "swv2_mlp_recompute_supported": (_I, [_I, _I]),
    "swv2_mlp_fwd": (_I, [C.POINTER(MlpArgs), _P]),
    "swv2_mlp_bwd_ws_floats": (C.c_size_t, [_I, _I]),
    "swv2_mlp_bwd": (_I, [C.POINTER(MlpBwdArgs), _P]),
    "swv2_cpb_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "swv2_cpb_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "swv2_cpb_bwd_ws": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, C.c_size_t, _P]),
    "swv2_cpb_bwd_ws_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "swv2_cpb_fwd_multi": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _F, _P]),
    "swv2_cpb_bwd_multi_ws_bytes": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "swv2_cpb_bwd_multi": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _F, _P, C.c_size_t, _P]),
    "swv2_block_fwd": (_I, [C.POINTER(BlockDesc), _P]),
    "swv2_block_bwd": (_I, [C.POINTER(BlockDesc), _P]),
}


def load() -> C.CDLL:
    """Load libswv2.so; raises Swv2Error when it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise Swv2Error(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the HIP library is required; there is no fallback path)")
            # torch first: the library then binds to the HIP runtime torch ships (same soname, already loaded).  Loaded the other way round the
            # process holds two runtimes and every launch of this library fails with "no ROCm-capable device is detected" (seen with build() and
            # smoke() in one process on a GPU box)
            import torch  # noqa: F401
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(lib, name)            # AttributeError if a declared symbol is missing
                fn.restype, fn.argtypes = res, args
            got = lib.swv2_version()
            if got != ABI_VERSION:         # a stale .so behind newer struct mirrors shifts every later member by one slot
                raise Swv2Error(f"{LIB_PATH} reports ABI revision {got}, this binding mirrors revision {ABI_VERSION} of include/swv2.h: "
                                "rebuild the library (python -c 'import __graft_entry__ as g; g.build()')")
            _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().swv2_last_error()
        raise Swv2Error(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
