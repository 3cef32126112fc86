"""Host-side mirror of the Moment-DETR set criterion (reference ``model/moment_detr.py:166-365``) over the device
matcher and ``uvtg_detr_criterion``: same constructor arguments, same ``forward(outputs, targets)`` contract, same
dictionary of losses (``loss_b, loss_g, loss_f, class_error, loss_s_intra, loss_contrastive_align`` and their
``_{i}`` copies for ``aux_outputs``).  Only the "l1" span loss is built -- it is the only one any script of the
reference selects.  There is no CPU path: the call fails loudly off-GPU."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .model import _f32c, _ptr, _stream

_NAMES = ("loss_b", "loss_g", "loss_f", "class_error", "loss_s_intra", "loss_contrastive_align")


class _DetrLosses(torch.autograd.Function):
    """losses[6] = f(pred_logits, pred_spans, saliency_scores, proj_queries, proj_txt_mem); the backward pass re-runs the kernel
    with the upstream gradient of each loss (the work is a few hundred KB, not worth stashing partial derivatives)."""

    @staticmethod
    def forward(ctx, crit, match, targets, logits, spans, sal, pq, pt):
        ctx.crit, ctx.match, ctx.targets = crit, match, targets
        ctx.save_for_backward(logits, spans, sal, pq, pt)
        return crit._launch(match, targets, logits, spans, sal, pq, pt, None)[0]

    @staticmethod
    def backward(ctx, go):
        logits, spans, sal, pq, pt = ctx.saved_tensors
        _, grads = ctx.crit._launch(ctx.match, ctx.targets, logits, spans, sal, pq, pt, _f32c(go))
        need = ctx.needs_input_grad[3:]
        return (None, None, None) + tuple(g if (n and g is not None) else None for g, n in zip(grads, need))


class SetCriterion(nn.Module):
    def __init__(self, matcher, weight_dict, eos_coef, losses, temperature, span_loss_type, max_v_l, saliency_margin=1):
        super().__init__()
        if span_loss_type != "l1":
            raise NotImplementedError("only the l1 span loss is implemented")
        self.matcher, self.weight_dict, self.losses = matcher, weight_dict, list(losses)
        self.temperature, self.span_loss_type, self.max_v_l = temperature, span_loss_type, max_v_l
        self.saliency_margin, self.eos_coef = saliency_margin, eos_coef
        self.foreground_label, self.background_label = 0, 1
        ew = torch.ones(2)
        ew[-1] = eos_coef
        self.register_buffer("empty_weight", ew)
        for name in self.losses:
            assert name in ("spans", "labels", "contrastive_align", "saliency"), f"do you really want to compute {name} loss?"

    def _launch(self, match, targets, logits, spans, sal, pq, pt, go):
        lib = _lib.load()
        op, ot, nm, tgt, off, max_t = match
        B, Q = spans.shape[:2]
        dev = logits.device
        has_sal, has_nce = sal.numel() > 0, pq.numel() > 0
        pos = neg = None
        n_pairs = L = T = D = 0
        if has_sal:
            pos = targets["saliency_pos_labels"].to(dev, torch.int64).contiguous()
            neg = targets["saliency_neg_labels"].to(dev, torch.int64).contiguous()
            n_pairs, L = pos.shape[1], sal.shape[1]
        if has_nce:
            T, D = pt.shape[1], pt.shape[2]
        out = torch.empty(6, device=dev)
        part = torch.empty(B, 8, device=dev)
        grads = [None] * 5
        if go is not None:
            grads = [torch.empty_like(logits), torch.empty_like(spans), torch.empty_like(sal) if has_sal else None,
                     torch.empty_like(pq) if has_nce else None, torch.empty_like(pt) if has_nce else None]
        nul = lambda t: _ptr(t) if t is not None else None
        _lib.check(lib.uvtg_detr_criterion(
            _ptr(logits), _ptr(spans), B, Q, _ptr(tgt), _ptr(off), _ptr(op), _ptr(ot), _ptr(nm), max_t,
            _ptr(sal) if has_sal else None, nul(pos), nul(neg), n_pairs, L, _ptr(pq) if has_nce else None,
            _ptr(pt) if has_nce else None, T, D, float(self.eos_coef), float(self.temperature), float(self.saliency_margin),
            nul(go), _ptr(part), _ptr(out), nul(grads[0]), nul(grads[1]), nul(grads[2]), nul(grads[3]), nul(grads[4]), _stream()),
            "uvtg_detr_criterion")
        return out, grads

    def _layer(self, outputs, targets, top):
        logits, spans = _f32c(outputs["pred_logits"]), _f32c(outputs["pred_spans"])
        if not logits.is_cuda:
            raise RuntimeError("univtg_amd SetCriterion runs on MI355X only (no CPU fallback)")
        if logits.shape[-1] != 2:
            raise ValueError("pred_logits must hold (foreground, background) logits")
        # device LSAP limits (uvtg_hungarian: 32 targets x 256 queries per sample).  Beyond them the kernel writes n_match = -1 and the
        # criterion would silently score the sample as all-background (HungarianMatcher.forward raises there): reject on the host,
        # from sizes the host already holds -- no device sync.
        sizes = [len(v["spans"]) for v in targets["span_labels"]]
        if (sizes and max(sizes) > 32) or spans.shape[1] > 256:
            raise RuntimeError(f"uvtg_hungarian: problem larger than the device LSAP limits (32 targets x 256 queries per sample); "
                               f"got max targets {max(sizes) if sizes else 0}, queries {spans.shape[1]}")
        match = self.matcher.match_device(outputs, targets)
        none = torch.empty(0, device=logits.device)
        use_sal = top and "saliency" in self.losses and "saliency_pos_labels" in targets
        sal = _f32c(outputs["saliency_scores"]) if use_sal else none
        use_nce = "contrastive_align" in self.losses
        pq = _f32c(outputs["proj_queries"]) if use_nce else none
        pt = _f32c(outputs["proj_txt_mem"]) if use_nce else none
        vec = _DetrLosses.apply(self, match, targets, logits, spans, sal, pq, pt)
        res = {}
        if "spans" in self.losses:
            res["loss_b"], res["loss_g"] = vec[0], vec[1]
        if "labels" in self.losses:
            res["loss_f"], res["class_error"] = vec[2], vec[3].detach()
        if top and "saliency" in self.losses:
            res["loss_s_intra"] = vec[4] if use_sal else 0
        if use_nce:
            res["loss_contrastive_align"] = vec[5]
        return res

    def forward(self, outputs, targets):
        top = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        losses = self._layer(top, targets, True)
        for i, aux in enumerate(outputs.get("aux_outputs", ())):
            losses.update({k + f"_{i}": v for k, v in self._layer(aux, targets, False).items()})
        return losses
