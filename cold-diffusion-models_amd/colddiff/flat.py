"""Flat parameter / gradient arenas and the fused optimizer tail.

All parameters of a model are re-homed into ONE contiguous fp32 buffer (and their gradients into a
second one), each tensor starting on a 16-byte boundary.  With 288 GB of HBM per MI355X there is no
reason to keep 238 separate allocations: Adam, EMA, zero_grad and the data-parallel gradient
all-reduce each become a single pass (or a few large buckets) over the arena at HBM / xGMI speed.
Replaces torch.optim.Adam + EMA of deblurring_diffusion_pytorch.py:68-81,1117,1200-1204.
"""
import torch

from . import runtime as rt
from .runtime import P


class FlatArena:
    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, "no parameters"
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            assert p.device == dev and p.dtype == torch.float32
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self._ptrs = []
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.detach().reshape(-1))
            old_grad = p.grad
            p.data = self.data[o:o + n].view(p.shape)
            g = self.grad[o:o + n].view(p.shape)
            if old_grad is not None:
                g.copy_(old_grad)
            p.grad = g
            self._ptrs.append(p.data_ptr())
        rt.bump_weights_epoch()

    def intact(self):
        """False if someone re-homed the parameters (e.g. module.cuda() after flattening)."""
        return all(p.data_ptr() == q and p.grad is not None and p.grad.data_ptr() == self.grad.data_ptr() + 4 * o
                   for p, q, o in zip(self.params, self._ptrs, self.offsets))

    def zero_grad(self):
        rt.lib().cdf_zero(P(self.grad), self.numel * 4, rt.stream(self.grad))

    def slice_of(self, p):
        i = next(k for k, q in enumerate(self.params) if q is p)
        return self.offsets[i], self.offsets[i] + p.numel()


class FusedAdam:
    """torch.optim.Adam(params, lr) with default betas/eps/no weight decay, as one kernel launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.arena = params if isinstance(params, FlatArena) else FlatArena(list(params))
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(self.arena.data)
        self.exp_avg_sq = torch.zeros_like(self.arena.data)
        self.step_count = 0
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "params": self.arena.params}]

    def step(self):
        a = self.arena
        assert a.intact(), "parameters were moved after the optimizer was built (call .cuda() before creating the Trainer)"
        self.step_count += 1
        g = self.param_groups[0]
        rt.lib().cdf_adam_step(P(a.data), P(a.grad), P(self.exp_avg), P(self.exp_avg_sq), a.numel, g["lr"], g["betas"][0], g["betas"][1],
                               g["eps"], self.step_count, rt.stream(a.data))
        rt.bump_weights_epoch()

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def ema_update(ema_arena, model_arena, beta):
    """EMA.update_model_average: ma = ma*beta + (1-beta)*p over every parameter (DEBLUR:73-81)."""
    assert ema_arena.numel == model_arena.numel
    rt.lib().cdf_ema_update(P(ema_arena.data), P(model_arena.data), ema_arena.numel, beta, rt.stream(ema_arena.data))
    rt.bump_weights_epoch()
