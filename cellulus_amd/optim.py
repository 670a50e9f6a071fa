"""Adam on libclx — the optimizer of ``cellulus/train.py:80-82``.

``torch.optim.Adam(params, lr, weight_decay=0.01)`` semantics (L2 coupled into
the gradient, bias correction, eps added after the sqrt), with the same
``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so
checkpoints written by either optimizer load into the other.  When the
parameters (and their gradients) are consecutive views of flat buffers — what
``UNetModel.flatten_parameters`` arranges — a step is ONE kernel launch.
"""

import torch
from torch.optim import Optimizer

from . import _clx


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("lr, eps and weight_decay must be non-negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid betas {betas}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False,
                        fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)

    @staticmethod
    def _consecutive(tensors):
        """True if the tensors are back-to-back slices of one allocation, in order."""
        p0 = tensors[0].data_ptr()
        off = 0
        for t in tensors:
            if not t.is_contiguous() or t.data_ptr() != p0 + 4 * off:
                return False
            off += t.numel()
        return True

    def _init_state(self, params):
        """exp_avg / exp_avg_sq live in flat buffers (one per group) with per-parameter views."""
        total = sum(p.numel() for p in params)
        dev = params[0].device
        m = _clx.zeros(total, torch.float32, dev)
        v = _clx.zeros(total, torch.float32, dev)
        off = 0
        for p in params:
            n = p.numel()
            st = self.state[p]
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = m[off:off + n].view(p.shape)
            st["exp_avg_sq"] = v[off:off + n].view(p.shape)
            off += n

    @torch.no_grad()
    def step(self, closure=None, guard=None):
        """`guard` (optional, not part of torch's signature): a one-element float64 device tensor; the update is
        enqueued at once and does nothing if the value is > 0 when the kernel runs (train._fused_step passes
        the loss kernel's bad-coordinate count, which the host reads only afterwards — `undo_step()` then takes
        the step counters back)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        guard_ptr = None
        self._advanced = []                      # what undo_step() may take back
        if guard is not None:
            if guard.dtype != torch.float64 or guard.numel() != 1:
                raise TypeError("guard must be a one-element float64 device tensor")
            _clx.require_device(guard, "guard")
            guard_ptr = _clx.ptr(guard)
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("decoupled_weight_decay"):
                raise NotImplementedError("cellulus_amd.optim.Adam: amsgrad/maximize/AdamW are not supported")
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                _clx.require_device(p, "parameter")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise TypeError("cellulus_amd.optim.Adam handles float32 parameters only")
            if any(len(self.state[p]) == 0 for p in params):
                self._init_state([p for p in params if len(self.state[p]) == 0])
            steps = set()
            for p in params:
                st = self.state[p]
                # torch.optim.Adam keeps `step` as a CPU float tensor; checkpoints written by older
                # torch versions hold a Python int — accept both, store back torch's layout
                if torch.is_tensor(st["step"]) and st["step"].device.type == "cpu" and st["step"].dtype == torch.float32:
                    st["step"] += 1                               # in place, as torch.optim.Adam does
                    n = int(st["step"].item())
                else:
                    n = (int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])) + 1
                    st["step"] = torch.tensor(float(n), dtype=torch.float32)
                steps.add(n)
            self._advanced.extend(params)
            b1, b2 = group["betas"]
            args = (float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                    float(group["weight_decay"]))
            stream = _clx.stream_ptr(params[0].device)
            grads = [p.grad for p in params]
            ms = [self.state[p]["exp_avg"] for p in params]
            vs = [self.state[p]["exp_avg_sq"] for p in params]
            if (len(steps) == 1 and self._consecutive(params) and self._consecutive(grads)
                    and self._consecutive(ms) and self._consecutive(vs)):
                n = sum(p.numel() for p in params)
                _clx.call("clx_adam_step_guarded", _clx.ptr(params[0]), _clx.ptr(grads[0]), _clx.ptr(ms[0]),
                          _clx.ptr(vs[0]), n, *args, steps.pop(), guard_ptr, stream)
            else:
                for p, g, m, v in zip(params, grads, ms, vs):
                    if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                        raise RuntimeError("cellulus_amd.optim.Adam needs contiguous parameters and gradients")
                    _clx.call("clx_adam_step_guarded", _clx.ptr(p), _clx.ptr(g), _clx.ptr(m), _clx.ptr(v),
                              p.numel(), *args, int(self.state[p]["step"].item()), guard_ptr, stream)
            # the kernels wrote behind torch's back: bump the autograd version counters so
            # cached packed weights (UNetModel) are refreshed
            for p in params:
                p.detach()[:0].zero_()
        return loss

    def undo_step(self):
        """Takes back the step COUNTERS of the last `step(guard=...)` whose guard turned out positive: the
        kernel left parameters and moments untouched, only the host-side counts had moved — and only those of
        the parameters that step advanced (a frozen or unused parameter, grad None, kept its count)."""
        for p in getattr(self, "_advanced", []):
            st = self.state[p]
            if int(st["step"].item()) > 0:
                st["step"] -= 1
        self._advanced = []
