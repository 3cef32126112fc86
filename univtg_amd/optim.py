"""`FusedAdamWClip`: the optimizer a maintainer swaps in when keeping the reference's own training loop.

The reference builds `torch.optim.AdamW([{"params": [p for n, p in model.named_parameters() if p.requires_grad]}], lr=opt.lr,
weight_decay=opt.wd)` (main/config.py:349-350) and runs `clip_grad_norm_(model.parameters(), opt.grad_clip)` + `optimizer.step()` per step
(main/train_vlp_ddp.py:65-68): on the drop-in model that is ~100 small launches over 174 MB of parameters.  This class keeps the
constructor shape, `zero_grad()` / `step()` / `state_dict()` / `load_state_dict()` and the `torch.optim.AdamW` checkpoint layout, and
runs clip + AdamW as the two kernels of `uvtg_adamw_clip_step` over ONE flat fp32 buffer (the table parameters are re-homed into it as
views: names, shapes and `state_dict()` of the model are unchanged).

    optimizer = FusedAdamWClip(param_dicts, lr=opt.lr, weight_decay=opt.wd, max_grad_norm=opt.grad_clip, model=model)

With `max_grad_norm > 0` the `clip_grad_norm_` line of the loop becomes redundant (leaving it in is harmless: a second clip of an already
clipped gradient is the identity).  Gradients are taken where autograd left them: the drop-in model's backward returns every parameter
gradient as a view of ONE flat buffer, which `AccumulateGrad` adopts as `p.grad` after `zero_grad()` (set_to_none) -- the kernels then
read that buffer in place; any other arrangement (accumulated gradients, DDP bucket views) is first gathered into the optimizer's own flat
gradient buffer by one `_foreach_copy_`.
"""
from __future__ import annotations

import torch

from . import _lib
from .model import Model, _ptr, _stream
from .trainer import flatten_parameters


class FusedAdamWClip(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0, model: Model | None = None):
        if model is None or not isinstance(model, Model):
            raise ValueError("FusedAdamWClip needs model=<the univtg_amd Model whose parameters it updates> (flat parameter layout)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamWClip keeps ONE parameter group (the reference's optimizer has one, main/config.py:349)")
        self.lib = _lib.load()
        self.model = model
        self.flat = getattr(model, "_flat", None)
        if self.flat is None:
            self.flat = flatten_parameters(model)           # p.data become views of one buffer; Parameter objects (and the group) are unchanged
        self.table = model._ordered_params()
        self.offs = model._offsets(model._dims(1, 4, 4, model.vid_dim, model.txt_dim, False))
        in_group = {id(p) for p in self.param_groups[0]["params"]}
        missing = [i for i, p in enumerate(self.table) if p.requires_grad and id(p) not in in_group]
        if missing:
            raise ValueError(f"{len(missing)} trainable table parameters are not in the optimizer's group: the fused update covers the whole flat buffer")
        if any(not p.requires_grad for p in self.table):
            raise ValueError("FusedAdamWClip updates every table parameter (frozen parameters are not supported)")
        dev = self.flat.device
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.gflat = None                                   # own gradient buffer, only for the gather path
        self.scratch = torch.zeros(1024, device=dev)        # UVTG_ADAMW_SCRATCH_FLOATS
        self.t = 0
        self._step_tensor = torch.tensor(0.0)
        self.in_place_steps = 0                             # steps that read autograd's flat gradient buffer in place (diagnostic)

    # ---- gradients ------------------------------------------------------------------------------------------------------------------
    def _grad_base(self):
        """Device address of a flat gradient buffer laid out like the parameters, if autograd's gradients ARE one (else None)."""
        g0 = self.table[0].grad
        if g0 is None:
            return None
        base = g0.data_ptr() - 4 * self.offs[0]
        store = g0.untyped_storage()
        if store.data_ptr() > base or store.data_ptr() + store.nbytes() < base + 4 * self.offs[-1]:
            return None
        for i, p in enumerate(self.table):
            g = p.grad
            if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * self.offs[i]
                    or g.untyped_storage().data_ptr() != store.data_ptr()):
                return None
        return base                                         # (the drop-in backward zeroes the alignment gaps between the parameters' ranges)

    def _gather(self):
        if self.gflat is None:
            self.gflat = torch.zeros_like(self.flat)
        dst, src = [], []
        for i, p in enumerate(self.table):
            view = self.gflat[self.offs[i]: self.offs[i] + p.numel()].view(p.shape)
            if p.grad is None:
                view.zero_()
            else:
                dst.append(view); src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        return self.gflat.data_ptr()

    # ---- the step -------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if all(p.grad is None for p in self.table):
            return loss
        g = self.param_groups[0]
        base = self._grad_base()
        if base is not None:
            self.in_place_steps += 1
        else:
            base = self._gather()
        self.t += 1
        _lib.check(self.lib.uvtg_adamw_clip_step(_ptr(self.flat), base, _ptr(self.m), _ptr(self.v), self.flat.numel(), float(g["lr"]),
                                                 float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.t,
                                                 float(g["max_grad_norm"]), 1.0, _ptr(self.scratch), _stream()), "uvtg_adamw_clip_step")
        self.model.invalidate_operand_cache()               # the raw kernel rewrote the parameters: model(...) rebuilds its MFMA operands
        self._publish_state()
        return loss

    def _publish_state(self):
        """`self.state` in torch.optim.AdamW's layout (views of the flat moment buffers; one shared step tensor)."""
        self._step_tensor.fill_(float(self.t))
        if self.state:
            return
        for i, p in enumerate(self.table):
            sl = slice(self.offs[i], self.offs[i] + p.numel())
            self.state[p] = {"step": self._step_tensor, "exp_avg": self.m[sl].view(p.shape), "exp_avg_sq": self.v[sl].view(p.shape)}

    def load_state_dict(self, state_dict):
        """Accepts a `torch.optim.AdamW` / `FusedAdamWClip` / `TrainStep` state dict; the moments are copied into the flat buffers."""
        super().load_state_dict(state_dict)
        steps = set()
        loaded = dict(self.state)
        self.state.clear()
        self.m.zero_(); self.v.zero_()
        for i, p in enumerate(self.table):
            ent = loaded.get(p)
            if not ent:
                continue
            sl = slice(self.offs[i], self.offs[i] + p.numel())
            self.m[sl].copy_(ent["exp_avg"].reshape(-1))
            self.v[sl].copy_(ent["exp_avg_sq"].reshape(-1))
            steps.add(int(float(ent["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamWClip keeps ONE step count for all parameters (the reference's optimizer steps them together)")
        self.t = steps.pop() if steps else 0
        if self.t:
            self._publish_state()
