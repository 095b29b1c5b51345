"""Adam on the depthcore kernel (csrc/adam.hip, `dc_adam_step`): the optimiser of the reference's training loop
(reference trainer.py:110-113 `optim.Adam(self.parameters_to_train, self.opt.learning_rate)`, `.step()` at trainer.py:238).

Drop-in for `torch.optim.Adam(params, lr, betas, eps)` on CUDA fp32 parameters: same arithmetic, same `state_dict()` layout
(`state[i] = {"step", "exp_avg", "exp_avg_sq"}`, torch's param_group keys), so `adam.pth` files interchange with the
reference's and `torch.optim.lr_scheduler.*` drive it unchanged.  The moments of all parameters live in two flat buffers
(the per-parameter `exp_avg` / `exp_avg_sq` are views), the step counts in a third; only the gradient addresses are
gathered per step.  No CPU path: parameters off the GPU raise DepthcoreError.
"""
import ctypes

import torch

from . import _lib
from ._lib import DepthcoreError, check


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters lr=%r betas=%r eps=%r" % (lr, betas, eps))
        # the keys torch.optim.Adam writes into its param_groups, so that state_dict()s interchange
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=True, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self._groups = None

    # ---- static tables -------------------------------------------------------------------------------------------------
    def _build(self):
        L = _lib.lib()
        chunk = L.dc_adam_chunk()
        self._groups = []
        for group in self.param_groups:
            if group["weight_decay"] or group["amsgrad"] or group["maximize"]:
                raise DepthcoreError("depthcore Adam covers the reference's configuration (no weight decay / amsgrad / maximize)")
            ps = [p for p in group["params"]]
            if not ps:
                self._groups.append(None)
                continue
            dev = ps[0].device
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                    raise DepthcoreError("depthcore Adam needs contiguous fp32 parameters on one GPU (got %s %s)" % (p.dtype, p.device))
            # 16-byte aligned slices of two flat moment buffers + one float of step count per tensor
            offs, n = [], 0
            for p in ps:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4
            m_flat = torch.zeros(n, dtype=torch.float32, device=dev)
            v_flat = torch.zeros(n, dtype=torch.float32, device=dev)
            steps = torch.zeros(len(ps), dtype=torch.float32, device=dev)
            slots = torch.empty(len(ps), 5, dtype=torch.int64)
            chunks, chunk_start = [], [0]
            for t, (p, off) in enumerate(zip(ps, offs)):
                st = self.state[p]
                old = dict(st)
                st["step"] = steps[t]
                st["exp_avg"] = m_flat[off:off + p.numel()].view_as(p)
                st["exp_avg_sq"] = v_flat[off:off + p.numel()].view_as(p)
                if old:                                    # state loaded before the first step (load_state_dict)
                    st["step"].copy_(torch.as_tensor(old["step"], dtype=torch.float32))
                    st["exp_avg"].copy_(old["exp_avg"])
                    st["exp_avg_sq"].copy_(old["exp_avg_sq"])
                slots[t, 0] = p.data_ptr()
                slots[t, 1] = st["exp_avg"].data_ptr()
                slots[t, 2] = st["exp_avg_sq"].data_ptr()
                slots[t, 3] = st["step"].data_ptr()
                slots[t, 4] = p.numel()
                chunks += [(t, s) for s in range(0, p.numel(), chunk)]
                chunk_start.append(len(chunks))
            self._groups.append(dict(
                params=ps, ptrs=[p.data_ptr() for p in ps], flat=(m_flat, v_flat, steps), slots=slots.to(dev),
                chunks=torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev),
                chunk_start=(ctypes.c_int * len(chunk_start))(*chunk_start), grads=(ctypes.c_void_p * len(ps))(),
                keep=[None] * len(ps)))

    def _stale(self):
        if self._groups is None or len(self._groups) != len(self.param_groups):
            return True
        for g, group in zip(self._groups, self.param_groups):
            ps = group["params"]
            if (g is None) != (not ps):
                return True
            if g is not None and (len(ps) != len(g["params"]) or any(a is not b or a.data_ptr() != q for a, b, q in zip(ps, g["params"], g["ptrs"]))):
                return True
        return False

    # ---- step ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._stale():
            self._build()
        L = _lib.lib()
        for g, group in zip(self._groups, self.param_groups):
            if g is None:
                continue
            grads, keep = g["grads"], g["keep"]
            dev = g["params"][0].device
            for t, p in enumerate(g["params"]):
                gr = p.grad
                if gr is None:
                    grads[t] = None
                    continue
                if gr.is_sparse or gr.dtype != torch.float32 or gr.device != dev:
                    raise DepthcoreError("depthcore Adam needs dense fp32 gradients on the parameters' GPU")
                if not gr.is_contiguous():
                    gr = gr.contiguous()
                keep[t] = gr                               # alive until the launch is enqueued (same stream: stream-ordered free)
                grads[t] = gr.data_ptr()
            b1, b2 = group["betas"]
            check(L.dc_adam_step(g["slots"].data_ptr(), g["chunks"].data_ptr(), g["chunk_start"], grads, len(g["params"]),
                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                 torch.cuda.current_stream(dev).cuda_stream), "dc_adam_step")
            for t in range(len(keep)):
                keep[t] = None
        return loss

    # ---- checkpoints -----------------------------------------------------------------------------------------------------
    def state_dict(self):
        """torch's layout, and the flags `torch.optim.Adam(params, lr)` would have written: `capturable` is how THIS class
        keeps its step counts (device floats), not a property of the checkpoint -- a reference-side `optim.Adam` that loads
        adam.pth must not inherit it (it would take torch's slower capturable path on the GPU and refuse CPU parameters).
        The step counts travel as CPU scalars, as torch's non-capturable Adam stores them."""
        sd = super().state_dict()
        groups = []
        for g in sd["param_groups"]:
            g = dict(g)
            g["capturable"], g["foreach"], g["fused"] = False, None, None
            groups.append(g)
        state = {}
        for k, st in sd["state"].items():
            st = dict(st)
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().to("cpu", torch.float32).clone()
            state[k] = st
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam's files (and its own).  The loaded moments are copied into the flat buffers at the next
        step (the base class replaces the state tensors; _build re-homes them)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:                   # files written by torch's Adam carry their own flags
            group["capturable"], group["fused"], group["foreach"] = True, None, None
        self._groups = None
