"""Drop-in replacement for the reference's ``model/univtg.py`` boundary on MI355X.

``build_model(args) -> (model, criterion)`` keeps the reference's factory signature
(model/univtg.py:409-450), ``Model.forward`` its call signature and output dict
(model/univtg.py:105-155), ``SetCriterion.forward`` its loss dict (model/univtg.py:338-351) and
``state_dict()`` the reference's checkpoint keys/shapes -- but every FLOP of the hot path
runs in the hand-written gfx950 kernels of ``libuvtg.so`` through the C ABI of ``include/uvtg.h``.
PyTorch only owns memory (parameters, activations, gradients), streams and autograd bookkeeping.

There is no CPU / eager fallback: tensors must live on a ROCm device and ``libuvtg.so`` must be built.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib

__all__ = ["build_model", "Model", "SetCriterion", "HungarianMatcher", "build_matcher"]


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _Bag(nn.Module):
    """Parameter container (no forward): only exists to reproduce the reference's state_dict keys."""


def _linear_bag(out_f, in_f):
    b = _Bag()
    b.weight = nn.Parameter(torch.empty(out_f, in_f))
    b.bias = nn.Parameter(torch.empty(out_f))
    nn.init.kaiming_uniform_(b.weight, a=math.sqrt(5))          # nn.Linear default (LinearLayer, univtg.py:392-395)
    bound = 1 / math.sqrt(in_f)
    nn.init.uniform_(b.bias, -bound, bound)
    return b


def _ln_bag(dim):
    b = _Bag()
    b.weight = nn.Parameter(torch.ones(dim))
    b.bias = nn.Parameter(torch.zeros(dim))
    return b


def _conv_bag(out_c, in_c, k=3):
    b = _Bag()
    b.weight = nn.Parameter(torch.empty(out_c, in_c, k))
    b.bias = nn.Parameter(torch.empty(out_c))
    nn.init.kaiming_uniform_(b.weight, a=math.sqrt(5))          # nn.Conv1d default (Conv, univtg.py:375-377)
    bound = 1 / math.sqrt(in_c * k)
    nn.init.uniform_(b.bias, -bound, bound)
    return b


def _proj_bag(in_f, hidden, n_layers):
    """input_{vid,txt}_proj: Sequential of LinearLayer(LayerNorm, net=[Dropout, Linear]) (univtg.py:91-100)."""
    seq = nn.ModuleList()
    for i in range(n_layers):
        blk = _Bag()
        blk.LayerNorm = _ln_bag(in_f if i == 0 else hidden)
        blk.net = nn.ModuleList([_Bag(), _linear_bag(hidden, in_f if i == 0 else hidden)])
        seq.append(blk)
    return seq


def _encoder_bag(d, F, E):
    tr = _Bag()
    tr.encoder = _Bag()
    layers = nn.ModuleList()
    for _ in range(E):
        lay = _Bag()
        lay.self_attn = _Bag()
        lay.self_attn.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        lay.self_attn.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        lay.self_attn.out_proj = _Bag()
        lay.self_attn.out_proj.weight = nn.Parameter(torch.empty(d, d))
        lay.self_attn.out_proj.bias = nn.Parameter(torch.zeros(d))
        lay.linear1 = _linear_bag(F, d)
        lay.linear2 = _linear_bag(d, F)
        lay.norm1 = _ln_bag(d)
        lay.norm2 = _ln_bag(d)
        layers.append(lay)
    tr.encoder.layers = layers
    for p in tr.parameters():                                   # Transformer._reset_parameters (droppath.py:32-35)
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return tr


class _UniVTGFunction(torch.autograd.Function):
    """Whole-model autograd node: one C call forward, one C call backward."""

    @staticmethod
    def forward(ctx, model, need_grad, src_txt, src_txt_mask, src_vid, src_vid_mask, *params):
        lib = _lib.load()
        B, Lv, Dv = src_vid.shape
        Lt, Dt = src_txt.shape[1], src_txt.shape[2]
        dev = src_vid.device
        training = bool(need_grad)                     # decided by the caller: Function.forward runs under no_grad
        if training and model.precision == "fp32x3":
            raise RuntimeError("backward is implemented for the bf16 arithmetic (precision='auto' or 'bf16'); 'fp32x3' is inference-only")
        dims = model._dims(B, Lv, Lt, Dv, Dt, training)
        ptrs = model._param_ptrs(params)
        wcache = model._prepare(dims, ptrs, params)
        lens = None
        if model.packed and not dims.precise and not model.return_memory and not model.use_txt_pos:
            # packed (ragged) encoder stream: the valid lengths come back from the masks (one device->host sync; the reference's
            # own loop synchronises every step too, main/train_vlp_ddp.py:71-73)
            hl = torch.stack([src_vid_mask.sum(1), src_txt_mask.sum(1)]).to(torch.int32).cpu().reshape(-1).tolist()
            lens = (C.c_int * (2 * B))(*hl)
        S, d = Lv + Lt, dims.d
        ws = torch.empty(lib.uvtg_workspace_bytes(C.byref(dims)), dtype=torch.uint8, device=dev)
        x0 = torch.empty(B, S, d, device=dev)
        pred_logits = torch.empty(B, Lv, 1, device=dev)
        pred_spans = torch.empty(B, Lv, 2, device=dev)
        txt_mem = torch.empty(B, 1, d, device=dev)
        sal = torch.empty(B, Lv, device=dev)
        memory = torch.empty(B, S, d, device=dev) if model.return_memory else None
        _lib.check(lib.uvtg_forward(C.byref(dims), ptrs, _ptr(wcache), _ptr(src_txt), _ptr(src_txt_mask), _ptr(src_vid),
                                    _ptr(src_vid_mask), _ptr(model._dim_t(dev)), _ptr(x0), _ptr(pred_logits), _ptr(pred_spans),
                                    _ptr(txt_mem), _ptr(sal), _ptr(memory), _ptr(ws), _stream(), lens), "uvtg_forward")
        ctx.model = None
        if training:
            ctx.model, ctx.dims, ctx.ws, ctx.wcache, ctx.lens = model, dims, ws, wcache, lens
            ctx.save_for_backward(src_txt, src_txt_mask, src_vid, src_vid_mask, x0, pred_logits, pred_spans, txt_mem, *params)
        # vid_mem_proj = x0[:, :Lv] leaves as an output of its own (a view): its gradient then arrives as the [B, Lv, d] tensor the criterion
        # wrote, not as a zero-filled [B, S, d] copy made by the slice's backward (190 MB of traffic per step at config 2); x0 itself is not
        # differentiable (nothing but that view is handed out), and absent upstream gradients stay None instead of materialised zeros
        vid_mem = x0[:, :Lv]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(x0, *([memory] if memory is not None else []))
        outs = (x0, pred_logits, pred_spans, txt_mem, sal, vid_mem)
        return outs + ((memory,) if memory is not None else ())

    @staticmethod
    def backward(ctx, g_x0, g_logits, g_spans, g_txt, g_sal, g_vid, *unused):
        lib = _lib.load()
        model, dims = ctx.model, ctx.dims
        src_txt, src_txt_mask, src_vid, src_vid_mask, x0, pred_logits, pred_spans, txt_mem, *params = ctx.saved_tensors
        ptrs = model._param_ptrs(params)
        offs = model._offsets(dims)
        grads = torch.empty(offs[-1], device=x0.device)
        gaps = model._offset_gaps(offs, params, x0.device)
        if gaps is not None:                           # alignment gaps between the parameters' ranges: FusedAdamWClip reads the buffer whole
            grads.index_fill_(0, gaps, 0.0)
        g = [None if t is None else _f32c(t) for t in (g_logits, g_spans, g_sal, g_txt, g_vid)]
        Lv, d = pred_logits.shape[1], x0.shape[2]
        _lib.check(lib.uvtg_backward(C.byref(dims), ptrs, _ptr(ctx.wcache), _ptr(src_txt), _ptr(src_txt_mask), _ptr(src_vid),
                                     _ptr(src_vid_mask), _ptr(x0), _ptr(pred_logits), _ptr(pred_spans), _ptr(txt_mem),
                                     _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]), _ptr(g[4]), Lv * d, d, None, None,
                                     _ptr(grads), _ptr(ctx.ws), _stream(), None, 0, ctx.lens), "uvtg_backward")
        ctx.ws = None
        out = [None] * 6
        for i, p in enumerate(params):
            out.append(grads[offs[i]: offs[i] + p.numel()].view_as(p) if p.requires_grad else None)
        return tuple(out)


class Model(nn.Module):
    """MI355X UniVTG model.  Same constructor surface as the reference's ``Model`` where it matters
    (hidden sizes instead of sub-modules) and the same ``forward`` signature / output dict."""

    def __init__(self, hidden_dim, nheads, dim_feedforward, enc_layers, txt_dim, vid_dim, input_dropout, dropout=0.1,
                 droppath=0.1, max_q_l=75, max_v_l=75, span_loss_type="l1", use_txt_pos=False, n_input_proj=2,
                 precision="auto", proj_precise=True, packed=False):
        super().__init__()
        if span_loss_type != "l1":
            raise NotImplementedError("span_loss_type='ce' is not implemented by the reference forward either (univtg.py:137-138)")
        if n_input_proj not in (1, 2, 3):
            raise ValueError("n_input_proj must be 1, 2 or 3 (model/univtg.py:89-100 builds at most three LinearLayer blocks)")
        # precision: "auto" (default) = the arithmetic follows the call: fp32x3 (fp16 hi+lo operand images, three MFMA products: ~22 bits, fp32-class) for calls under
        #                 torch.no_grad() -- the inference path of main/inference_mr.py:88-193, where north_star asks for span indices
        #                 after NMS identical to the fp32 reference -- and bf16 MFMA operands when a backward will follow (training);
        #            "bf16"   = bf16 operands for every call (fast inference, opt-in: post-NMS top-1 agrees with fp32 for ~98 % of the
        #                 samples, the ordered top-10 list for ~85 %, tests/test_gpu_parity_full.py);
        #            "fp32x3" = fp32-class arithmetic, inference only.
        if precision not in ("auto", "bf16", "fp32x3"):
            raise ValueError("precision must be 'auto', 'bf16' or 'fp32x3'")
        d = hidden_dim
        self.hidden_dim, self.nheads, self.dim_feedforward, self.enc_layers = d, nheads, dim_feedforward, enc_layers
        self.txt_dim, self.vid_dim = txt_dim, vid_dim
        self.input_dropout, self.dropout, self.droppath = float(input_dropout), float(dropout), float(droppath)
        self.span_loss_type, self.max_v_l, self.use_txt_pos, self.n_input_proj = span_loss_type, max_v_l, bool(use_txt_pos), n_input_proj
        self.max_q_l = max_q_l
        # proj_precise: True (default) = input projections always on split (fp16 hi+lo) operands: saliency_scores -- a function of the projections
        #                       alone -- stay within 1e-4 of the fp32 reference in inference AND under training dropout (measured 2.5e-7; +3.4 %
        #                       step time at config 2, the backward reads a bf16 copy of the operand),
        #               False = always plain bf16 operands (saliency within 3e-2 in train mode),
        #               "auto" = split operands for inference calls (no gradient), plain bf16 when a backward will follow
        if proj_precise not in (True, False, "auto"):
            raise ValueError("proj_precise must be True, False or 'auto'")
        self.precision, self.proj_precise, self.return_memory = precision, proj_precise, False
        # packed=True: run the encoder on the packed (ragged) row stream -- identical results, fewer rows on ragged batches; costs one
        # device->host read of the mask sums per call.  The engine picks the exact variant itself (include/uvtg.h, lens_host): valid
        # rows + one representative padded clip per sample, or -- under training-time input / attention dropout, where every padded
        # clip carries its own mask -- all clip rows and the valid text tokens
        self.packed = bool(packed)
        # ---- parameters, registered in the reference's order / names ----
        self.transformer = _encoder_bag(d, dim_feedforward, enc_layers)
        self.txt_position_embed = _Bag()                                    # read only with use_txt_pos (always kept for ckpt parity)
        self.txt_position_embed.position_embeddings = _Bag()
        self.txt_position_embed.position_embeddings.weight = nn.Parameter(torch.randn(max_q_l, d))
        self.txt_position_embed.LayerNorm = _ln_bag(d)
        self.token_type_embeddings = _Bag()
        self.token_type_embeddings.weight = nn.Parameter(torch.empty(2, d).normal_(0.0, 0.02))
        for name, out in (("span_embed", 2), ("class_embed", 1)):
            head = _Bag()
            head.layers = nn.ModuleList([_conv_bag(d, d), _conv_bag(d, d), _conv_bag(out, d)])
            setattr(self, name, head)
        self.input_txt_proj = _proj_bag(txt_dim, d, n_input_proj)
        self.input_vid_proj = _proj_bag(vid_dim, d, n_input_proj)
        self.weightedpool = _Bag()
        self.weightedpool.weight = nn.Parameter(nn.init.xavier_uniform_(torch.empty(d, 1)))
        self._step = 0
        self._seed = 0x5EED
        self._wcache = {}
        self._param_epoch = 0
        self._dimt = None
        self._off_cache = {}

    # ---- parameter table in the C-ABI order (include/uvtg.h) ----
    def _ordered_params(self):
        ps = []
        for lay in self.transformer.encoder.layers:
            ps += [lay.self_attn.in_proj_weight, lay.self_attn.in_proj_bias, lay.self_attn.out_proj.weight,
                   lay.self_attn.out_proj.bias, lay.linear1.weight, lay.linear1.bias, lay.linear2.weight, lay.linear2.bias,
                   lay.norm1.weight, lay.norm1.bias, lay.norm2.weight, lay.norm2.bias]
        ps.append(self.token_type_embeddings.weight)
        for head in (self.span_embed, self.class_embed):
            for c in head.layers:
                ps += [c.weight, c.bias]
        for proj in (self.input_txt_proj, self.input_vid_proj):
            for blk in proj:
                ps += [blk.LayerNorm.weight, blk.LayerNorm.bias, blk.net[1].weight, blk.net[1].bias]
        ps.append(self.weightedpool.weight)
        if self.use_txt_pos:      # --use_txt_pos (model/univtg.py:123): the trainable text positions join the table (and get gradients)
            ps += [self.txt_position_embed.position_embeddings.weight, self.txt_position_embed.LayerNorm.weight,
                   self.txt_position_embed.LayerNorm.bias]
        return ps

    def _dims(self, B, Lv, Lt, Dv, Dt, training):
        if Dv != self.vid_dim or Dt != self.txt_dim:
            raise ValueError(f"feature dims ({Dv}, {Dt}) do not match the model ({self.vid_dim}, {self.txt_dim})")
        if training:
            self._step += 1
        return _lib.Dims(B=B, Lv=Lv, Lt=Lt, d=self.hidden_dim, H=self.nheads, F=self.dim_feedforward, E=self.enc_layers,
                         Dv=Dv, Dt=Dt, n_proj=self.n_input_proj,
                         precise=int(self.precision == "fp32x3" or (self.precision == "auto" and not training)),
                         training=int(training), proj_precise=int((not training) if self.proj_precise == "auto" else bool(self.proj_precise)),
                         p_in=self.input_dropout if self.training else 0.0,
                         p_attn=self.dropout if self.training else 0.0,
                         p_path=self.droppath if self.training else 0.0,
                         seed=(self._seed * 1000003 + self._step) & 0xFFFFFFFFFFFFFFFF,
                         use_txt_pos=int(self.use_txt_pos), max_q_l=self.max_q_l)

    def _param_ptrs(self, params):
        arr = (C.c_void_p * len(params))()
        for i, p in enumerate(params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous fp32 tensors")
            arr[i] = p.data_ptr()
        return arr

    def _offsets(self, dims):
        key = (dims.d, dims.F, dims.E, dims.Dv, dims.Dt, dims.n_proj, dims.use_txt_pos, dims.max_q_l)
        if key not in self._off_cache:
            lib = _lib.load()
            n = lib.uvtg_param_count(C.byref(dims))
            buf = (C.c_longlong * (n + 1))()
            _lib.check(lib.uvtg_param_offsets(C.byref(dims), buf), "uvtg_param_offsets")
            self._off_cache[key] = list(buf)
        return self._off_cache[key]

    def _offset_gaps(self, offs, params, dev):
        """int64 indices of the flat layout's alignment gaps (every parameter starts at a multiple of 4 elements), or None."""
        key = ("gaps", tuple(offs), str(dev))
        if key not in self._off_cache:
            idx = [j for i, p in enumerate(params) for j in range(offs[i] + p.numel(), offs[i + 1])]
            self._off_cache[key] = torch.tensor(idx, dtype=torch.long, device=dev) if idx else None
        return self._off_cache[key]

    def _prepare(self, dims, ptrs, params):
        """(Re)build the MFMA operand cache when any parameter changed (tracked by tensor versions)."""
        lib = _lib.load()
        key = (dims.precise, dims.training, dims.proj_precise, dims.Dv, dims.Dt)
        # _param_epoch: bumped by whoever rewrites the parameter storage behind autograd's back (TrainStep's raw AdamW kernel
        # updates the flat buffer the parameters are views of: tensor versions do not move)
        sig = (self._param_epoch,) + tuple((p.data_ptr(), p._version) for p in params)
        ent = self._wcache.get(key)
        if ent is None or ent[0] != sig:
            nbytes = lib.uvtg_wcache_bytes(C.byref(dims))
            buf = ent[1] if ent is not None and ent[1].numel() == nbytes else torch.empty(nbytes, dtype=torch.uint8, device=params[0].device)
            _lib.check(lib.uvtg_prepare_weights(C.byref(dims), ptrs, _ptr(buf), _stream()), "uvtg_prepare_weights")
            self._wcache[key] = (sig, buf)
            ent = self._wcache[key]
        return ent[1]

    def _dim_t(self, dev):
        if self._dimt is None or self._dimt.device != dev:
            d = self.hidden_dim                                       # position_encoding.py:75-78, evaluated by torch on host
            i = torch.arange(d, dtype=torch.float32)
            self._dimt = (10000 ** (2 * torch.div(i, 2).int() / d)).to(dev)
        return self._dimt

    def set_seed(self, seed: int):
        self._seed, self._step = int(seed), 0

    def invalidate_operand_cache(self):
        """Call after rewriting parameter storage without going through torch in-place ops (raw kernels on a flat buffer)."""
        self._param_epoch += 1

    def forward(self, src_txt, src_txt_mask, src_vid, src_vid_mask, src_cls=None, src_cls_mask=None):
        if not src_vid.is_cuda:
            raise RuntimeError("univtg_amd runs on MI355X only: inputs must be on a ROCm device (no CPU fallback)")
        params = self._ordered_params()
        args = [_f32c(t) for t in (src_txt, src_txt_mask, src_vid, src_vid_mask)]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        res = _UniVTGFunction.apply(self, need_grad, *args, *params)
        x0, pred_logits, pred_spans, txt_mem, sal, vid_mem = res[:6]
        out = {"pred_logits": pred_logits, "pred_spans": pred_spans, "src_vid_mask": src_vid_mask,
               "vid_mem_proj": vid_mem, "txt_mem_proj": txt_mem, "saliency_scores": sal}
        if self.return_memory:
            out["memory"] = res[6]
        if src_cls is not None:
            # TAL pre-training branch (model/univtg.py:109-117,151-153): the class-name token features take the TEXT projection, the text type
            # embedding and the weighted pool -- exactly what the engine computes as txt_mem_proj.  They never enter the encoder, so they run as a
            # second engine call whose "queries" are the class names (one dummy clip per class; its other outputs are dropped, and under autograd
            # its backward hands the text projection / type embedding / pool their gradient through the same path as the real queries')
            n_cls = src_cls.shape[0]
            if src_cls_mask is None:
                raise ValueError("src_cls needs src_cls_mask")
            dummy_vid = torch.zeros(n_cls, 1, self.vid_dim, device=src_vid.device)
            dummy_mask = torch.ones(n_cls, 1, device=src_vid.device)
            was_mem, self.return_memory = self.return_memory, False
            try:
                cres = _UniVTGFunction.apply(self, need_grad, _f32c(src_cls), _f32c(src_cls_mask), dummy_vid, dummy_mask, *params)
            finally:
                self.return_memory = was_mem
            out["cls_mem_proj"] = cres[3][:, 0]
        return out


class _CriterionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, which, eos_coef, pred_logits, pred_spans, vid, txt, timestamp, ts_mask, ts_window, span_nn, sal, pos_idx):
        lib = _lib.load()
        B, Lv = pred_logits.shape[0], pred_logits.shape[1]
        d = vid.shape[-1]
        if vid.stride(2) != 1:
            vid = vid.contiguous()
        pl, ps, tx = _f32c(pred_logits), _f32c(pred_spans), _f32c(txt)
        ws = torch.empty(lib.uvtg_loss_ws_floats(B, Lv, d), device=pl.device)
        losses = torch.empty(8, device=pl.device)
        _lib.check(lib.uvtg_criterion_fwd(B, Lv, d, which, eos_coef, _ptr(pl), _ptr(ps), _ptr(vid), vid.stride(0), vid.stride(1),
                                          _ptr(tx), _ptr(timestamp), _ptr(ts_mask), _ptr(ts_window), _ptr(span_nn), _ptr(sal),
                                          _ptr(pos_idx), _ptr(ws), _ptr(losses), None, None, None, _stream()), "uvtg_criterion_fwd")
        ctx.which, ctx.eos, ctx.ws = which, eos_coef, ws
        ctx.shapes = (pred_logits.shape, pred_spans.shape, txt.shape)
        ctx.save_for_backward(pl, ps, vid, tx, timestamp, ts_mask, ts_window, span_nn, sal, pos_idx, losses)
        return losses[:6].clone()          # five losses + the device-side "saliency terms active" flag (the saliency_scores.sum() == 0 early-out)

    @staticmethod
    def backward(ctx, go):
        lib = _lib.load()
        pl, ps, vid, tx, timestamp, ts_mask, ts_window, span_nn, sal, pos_idx, losses = ctx.saved_tensors
        B, Lv, d = pl.shape[0], pl.shape[1], vid.shape[-1]
        go = _f32c(go)
        g_l, g_s = torch.empty(B, Lv, device=pl.device), torch.empty(B, Lv, 2, device=pl.device)
        g_v, g_t = torch.empty(B, Lv, d, device=pl.device), torch.empty(B, d, device=pl.device)
        g_c, g_r = torch.empty(B, Lv, device=pl.device), torch.empty(B, d, device=pl.device)
        _lib.check(lib.uvtg_criterion_bwd(B, Lv, d, ctx.which, ctx.eos, _ptr(pl), _ptr(ps), _ptr(vid), vid.stride(0), vid.stride(1),
                                          _ptr(tx), _ptr(timestamp), _ptr(ts_mask), _ptr(ts_window), _ptr(span_nn), _ptr(sal),
                                          _ptr(pos_idx), _ptr(ctx.ws), _ptr(losses), _ptr(go), _ptr(g_l), _ptr(g_s), _ptr(g_v),
                                          _ptr(g_t), _ptr(g_c), _ptr(g_r), None, None, None, _stream()), "uvtg_criterion_bwd")
        sl, ss, st = ctx.shapes
        return (None, None, g_l.view(sl), g_s.view(ss), g_v, g_t.view(st)) + (None,) * 6


class _ClsNceFunction(torch.autograd.Function):
    """Class term of the 'saliency_cls' loss (model/univtg.py:314-324) on device: uvtg_cls_nce_fwd / _bwd."""

    @staticmethod
    def forward(ctx, vid, cls, pos_idx, cls_idx, active):
        lib = _lib.load()
        B, Lv, d = vid.shape
        C_ = cls.shape[0]
        if vid.stride(2) != 1:
            vid = vid.contiguous()
        cls, cls_idx = _f32c(cls), _f32c(cls_idx)
        ws = torch.empty(lib.uvtg_cls_nce_ws_floats(B, C_), device=vid.device)
        loss = torch.empty(1, device=vid.device)
        _lib.check(lib.uvtg_cls_nce_fwd(B, C_, d, _ptr(vid), vid.stride(0), vid.stride(1), _ptr(pos_idx), _ptr(cls), _ptr(cls_idx), _ptr(active),
                                        _ptr(ws), _ptr(loss), _stream()), "uvtg_cls_nce_fwd")
        ctx.save_for_backward(vid, cls, pos_idx, cls_idx, active, ws)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        lib = _lib.load()
        vid, cls, pos_idx, cls_idx, active, ws = ctx.saved_tensors
        B, Lv, d = vid.shape
        C_ = cls.shape[0]
        go = _f32c(go).reshape(1)
        g_vid = torch.zeros(B, Lv, d, device=vid.device)          # (only row pos_idx[b] of every sample is non-zero)
        g_cls = torch.empty(C_, d, device=vid.device)
        _lib.check(lib.uvtg_cls_nce_bwd(B, C_, d, _ptr(vid), vid.stride(0), vid.stride(1), _ptr(pos_idx), _ptr(cls), _ptr(cls_idx), _ptr(active),
                                        _ptr(ws), _ptr(go), _ptr(g_vid), Lv * d, d, _ptr(g_cls), _stream()), "uvtg_cls_nce_bwd")
        return g_vid, g_cls, None, None, None


class SetCriterion(nn.Module):
    """Dense UniVTG criterion (model/univtg.py:157-351) on device; same constructor, ``weight_dict`` and loss keys."""

    def __init__(self, matcher, weight_dict, eos_coef, losses, temperature, span_loss_type, max_v_l, saliency_margin=1):
        super().__init__()
        self.matcher, self.weight_dict, self.losses = matcher, weight_dict, losses
        self.span_loss_type, self.max_v_l, self.saliency_margin = span_loss_type, max_v_l, saliency_margin
        self.temperature = 0.07                       # the reference overwrites --temperature (univtg.py:185)
        self.foreground_label, self.background_label, self.eos_coef = 0, 1, eos_coef
        empty_weight = torch.ones(2)
        empty_weight[-1] = eos_coef
        self.register_buffer("empty_weight", empty_weight)
        for name in losses:
            if name not in ("spans", "labels", "saliency", "saliency_cls"):
                raise NotImplementedError(f"loss '{name}' is outside the accelerated path (spans / labels / saliency / saliency_cls)")
        if "saliency" in losses and "saliency_cls" in losses:
            raise ValueError("'saliency' and 'saliency_cls' write the same loss keys (the reference's factory selects one, model/univtg.py:436-438)")

    def forward(self, outputs, targets, hl_only=False):
        cls_mode = "saliency_cls" in self.losses
        which = (1 if "spans" in self.losses else 0) | (2 if "labels" in self.losses else 0) | (4 if ("saliency" in self.losses or cls_mode) else 0)
        pl = outputs["pred_logits"]
        if not pl.is_cuda:
            raise RuntimeError("univtg_amd criterion runs on MI355X only (no CPU fallback)")
        f = lambda k: _f32c(targets[k])
        has_sal = ("saliency_pos_labels" in targets) and ("saliency_scores" in targets)
        sal = f("saliency_scores") if has_sal else None
        pos = targets["saliency_pos_labels"][:, 0].long().contiguous() if has_sal else None
        res = _CriterionFunction.apply(which, float(self.eos_coef), pl, outputs["pred_spans"], outputs["vid_mem_proj"],
                                       outputs["txt_mem_proj"], f("timestamp"), f("timestamp_mask"), f("timestamp_window"),
                                       f("span_labels_nn"), sal, pos)
        out = {}
        if which & 1:
            out["loss_b"], out["loss_g"] = res[0], res[1]
        if which & 2:
            out["loss_f"] = res[2]
        if which & 4 and not cls_mode:
            out["loss_s_inter"], out["loss_s_intra"] = res[3], res[4]
        elif cls_mode:
            # 'saliency_cls' (model/univtg.py:284-326): the inter-video term is loss_saliency's (the intra-video term the kernel computes beside
            # it is not used: no gradient flows into it); the class term only with targets['cls_idx'] (absent in evaluation, :312-313).  Both
            # early-outs (:286-290) come out as zero tensors without a host synchronisation: no positive labels -> the kernel is given no
            # saliency targets; saliency_scores.sum() == 0 -> the device-side flag res[5] zeroes value and gradient
            out["loss_s_inter"] = res[3]
            if "cls_idx" in targets or not has_sal:
                if has_sal:
                    out["loss_s_intra"] = _ClsNceFunction.apply(outputs["vid_mem_proj"], outputs["cls_mem_proj"], pos, targets["cls_idx"], res[5:6].detach())
                else:
                    out["loss_s_intra"] = res[4] * 0.0
        return out


class HungarianMatcher(nn.Module):
    """model/matcher.py:13-100 on device (cost matrix + per-sample LSAP kernels); same call signature/return."""

    def __init__(self, cost_class=1.0, cost_span=1.0, cost_giou=1.0, span_loss_type="l1", max_v_l=75):
        super().__init__()
        if span_loss_type != "l1":
            raise NotImplementedError("only the l1 span cost is implemented")
        assert cost_class != 0 or cost_span != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_span, self.cost_giou = cost_class, cost_span, cost_giou
        self.span_loss_type, self.max_v_l, self.foreground_label = span_loss_type, max_v_l, 0

    @torch.no_grad()
    def match_device(self, outputs, targets):
        """The matching as device arrays (no host sync): (pred_idx [B,max_t] int64, tgt_idx, n_match [B] int32, tgt_cxw [T,2],
        tgt_off [B+1] int32, max_t) -- the layout uvtg_detr_criterion consumes."""
        lib = _lib.load()
        logits, spans = _f32c(outputs["pred_logits"].detach()), _f32c(outputs["pred_spans"].detach())
        if not logits.is_cuda:
            raise RuntimeError("univtg_amd matcher runs on MI355X only (no CPU fallback)")
        B, Q = spans.shape[:2]
        tg = targets["span_labels"]
        sizes = [len(v["spans"]) for v in tg]
        max_t = max(1, max(sizes))
        dev = logits.device
        tgt = torch.cat([_f32c(v["spans"]).to(dev) for v in tg]) if sum(sizes) else torch.zeros(1, 2, device=dev)
        off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0).tolist()), dtype=torch.int32, device=dev)
        cost = torch.empty(B, Q, max_t, device=dev)
        op = torch.empty(B, max_t, dtype=torch.int64, device=dev)
        ot = torch.empty(B, max_t, dtype=torch.int64, device=dev)
        nm = torch.empty(B, dtype=torch.int32, device=dev)
        _lib.check(lib.uvtg_hungarian(_ptr(logits), logits.shape[-1], _ptr(spans), B, Q, _ptr(tgt), _ptr(off), max_t,
                                      float(self.cost_class), float(self.cost_span), float(self.cost_giou),
                                      _ptr(cost), _ptr(op), _ptr(ot), _ptr(nm), _stream()), "uvtg_hungarian")
        return op, ot, nm, tgt, off, max_t

    @torch.no_grad()
    def forward(self, outputs, targets):
        op, ot, nm = self.match_device(outputs, targets)[:3]
        B = op.shape[0]
        op, ot, nm = op.cpu(), ot.cpu(), nm.cpu().tolist()
        if any(n < 0 for n in nm):
            raise RuntimeError("uvtg_hungarian: problem larger than the device LSAP limits (32 x 256)")
        return [(op[b, :nm[b]].clone(), ot[b, :nm[b]].clone()) for b in range(B)]


def build_matcher(args):
    return HungarianMatcher(cost_span=args.set_cost_span, cost_giou=args.set_cost_giou, cost_class=args.set_cost_class,
                            span_loss_type=args.span_loss_type, max_v_l=args.max_v_l)


def build_model(args):
    """Same contract as the reference factory (model/univtg.py:409-450): reads the same ``args`` fields and
    returns ``(model, criterion)``; ``main/config.py:setup_model`` moves the model to the device itself."""
    device = torch.device(args.device)
    model = Model(hidden_dim=args.hidden_dim, nheads=args.nheads, dim_feedforward=args.dim_feedforward,
                  enc_layers=args.enc_layers, txt_dim=args.t_feat_dim, vid_dim=args.v_feat_dim,
                  input_dropout=args.input_dropout, dropout=args.dropout, droppath=args.droppath,
                  max_q_l=args.max_q_l, max_v_l=getattr(args, "max_v_l", 75), span_loss_type=args.span_loss_type,
                  use_txt_pos=args.use_txt_pos, n_input_proj=args.n_input_proj,
                  precision=getattr(args, "precision", "auto"), proj_precise=getattr(args, "proj_precise", True), packed=getattr(args, "packed", False))
    if getattr(args, "pre_norm", False):
        raise NotImplementedError("--pre_norm crashes in the reference too (forward_pre is undefined, droppath.py:133)")
    matcher = build_matcher(args)
    weight_dict = {"loss_b": args.b_loss_coef, "loss_g": args.g_loss_coef, "loss_f": args.f_loss_coef,
                   "loss_s_intra": args.s_loss_intra_coef, "loss_s_inter": args.s_loss_inter_coef}
    if args.dset_type in ["mr", "vlp"]:
        if "tal" not in args.train_path:
            losses = ["spans", "labels", "saliency"]
        else:
            losses = ["spans", "labels", "saliency_cls"]      # model/univtg.py:436-438 (needs src_cls / src_cls_mask in the model call)
    elif args.dset_type in ["hl", "vs"]:
        losses = ["labels", "saliency"]
    else:
        raise ValueError(args.dset_type)
    criterion = SetCriterion(matcher=matcher, weight_dict=weight_dict, losses=losses, eos_coef=args.eos_coef,
                             temperature=args.temperature, span_loss_type=args.span_loss_type, max_v_l=args.max_v_l,
                             saliency_margin=args.saliency_margin)
    criterion.to(device)
    return model, criterion
