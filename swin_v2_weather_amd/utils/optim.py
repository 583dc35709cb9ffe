"""Adam on the HIP multi-tensor kernel (reference train.py:176: `optim.Adam(model.parameters(), lr, betas=(0.9, 0.95))`,
stepped at train.py:330).

`HipAdam` is a drop-in `torch.optim.Adam`: same constructor, same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq`
per parameter, so the reference's checkpoints restore into it and its checkpoints restore into torch's Adam).  `step()`
updates every fp32 CUDA parameter of a group with ONE `swv2_adam_multi` launch (torch's fused path: 5 launches of ~45 us
for the depth-12 model); options the kernel does not implement (amsgrad, weight decay, maximize, non-fp32 or CPU
parameters) take torch's own path for that group, so nothing is silently dropped.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib as L


class _Item(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_long)]


def _dense(t: torch.Tensor) -> bool:
    """memory is one gap-free block (the kernel walks it linearly; p, grad, m, v must then only share the strides)"""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) or \
        (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d))


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("fused", None)
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._tables = {}       # group index -> cached launch tables

    # ---- launch tables -------------------------------------------------------------------------------------
    def _table(self, gi, plist):
        """(items host array, pinned staging, device items, device chunks, n_chunks); sizes are static per group, pointers
        are refreshed by `_refresh` (gradient buffers may move between steps)."""
        key = tuple((p.data_ptr(), p.numel()) for p in plist)
        t = self._tables.get(gi)
        if t is not None and t["key"] == key:
            return t
        dev = plist[0].device
        chunk = L.load().swv2_adam_chunk()
        pairs = []
        for i, p in enumerate(plist):
            pairs += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        chunks = torch.tensor(pairs, dtype=torch.int32).to(dev)
        nbytes = C.sizeof(_Item) * len(plist)
        t = {"key": key, "host": [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)],
             "dev": torch.empty(nbytes, dtype=torch.uint8, device=dev), "chunks": chunks, "n_chunks": len(pairs), "ptrs": None}
        self._tables[gi] = t
        return t

    @staticmethod
    def _refresh(t, ptrs, n):
        """upload the pointer table.  Two pinned staging buffers are used alternately, each guarded by an event recorded behind
        its H2D copy: the host may run several steps ahead of the GPU (no per-step sync in bench.py / train.py), so a staging
        buffer is rewritten only after the copy that last read it has executed (without the wait, step N's table
        could be overwritten with step N + 2's pointers before step N's copy ran)."""
        t["flip"] = 1 - t.get("flip", 0)
        host = t["host"][t["flip"]]
        ev = t.setdefault("events", [None, None])
        if ev[t["flip"]] is not None:
            ev[t["flip"]].synchronize()
        np.frombuffer(host.numpy(), dtype=np.int64).reshape(n, 5)[:] = np.asarray(ptrs, dtype=np.int64)
        t["dev"].copy_(host, non_blocking=True)
        e = torch.cuda.Event()
        e.record(torch.cuda.current_stream(t["dev"].device))
        ev[t["flip"]] = e
        t["ptrs"] = ptrs

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}

    # ---- step ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, grad_inv_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        slow = []
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if (group["amsgrad"] or group["weight_decay"] != 0 or group["maximize"] or isinstance(group["lr"], torch.Tensor) or
                    not plist):
                slow.append(gi)
                continue
            states = []
            for p in plist:
                s = self.state[p]
                if len(s) == 0:
                    s["step"] = torch.tensor(0.0, dtype=torch.float32)
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                states.append(s)
            t = self._table(gi, plist)
            ptrs = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), p.numel())
                         for p, s in zip(plist, states))
            if ptrs != t["ptrs"]:
                # new tensors: the kernel walks raw memory, so every (p, grad, m, v) must be fp32 on the GPU, dense and laid
                # out alike, and the per-parameter step counters must agree; anything else keeps torch's exact path
                ok = all(p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse and
                         _dense(p) and p.grad.stride() == p.stride() and
                         s["exp_avg"].stride() == p.stride() and s["exp_avg_sq"].stride() == p.stride()
                         for p, s in zip(plist, states))
                if ok and t.get("step") is None:           # new table (first step, new parameters, restored state)
                    steps = {int(s["step"]) for s in states}
                    ok = len(steps) == 1
                    if ok:
                        t["step"], t["step_tensors"] = steps.pop(), [s["step"] for s in states]
                if not ok:
                    self._tables.pop(gi, None)
                    slow.append(gi)
                    continue
                self._refresh(t, ptrs, len(plist))
            t["step"] += 1
            step = t["step"]
            stream = torch.cuda.current_stream(plist[0].device).cuda_stream
            b1, b2 = group["betas"]
            L.check(L.load().swv2_adam_multi(t["dev"].data_ptr(), t["chunks"].data_ptr(), t["n_chunks"], float(group["lr"]),
                                             float(b1), float(b2), float(group["eps"]), step, float(grad_inv_scale), stream),
                    "swv2_adam_multi")
            torch._foreach_add_(t["step_tensors"], 1.0)      # state_dict() keeps torch's per-parameter `step` entries
            # the kernel wrote the parameters behind autograd's back: bump their version counters like an in-place torch op
            # would (anything keyed on `Tensor._version` -- e.g. cached casts of the weights -- must see the change)
            torch.autograd.graph.increment_version(plist)
        if slow:
            keep = self.param_groups
            try:
                self.param_groups = [keep[i] for i in slow]
                if grad_inv_scale != 1.0:           # torch's path knows no gradient scale: unscale these groups' gradients first
                    gs = [p.grad for g_ in self.param_groups for p in g_["params"] if p.grad is not None]
                    if gs:
                        torch._foreach_mul_(gs, float(grad_inv_scale))
                super().step()
            finally:
                self.param_groups = keep
        return loss
