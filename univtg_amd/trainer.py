"""Native data-parallel training step for the UniVTG hot path on MI355X.

One step = forward -> criterion -> backward -> (RCCL all-reduce over xGMI) -> global-norm clip + AdamW, i.e. the
body of ``train_epoch`` in the reference (main/train_vlp_ddp.py:44-75), issued as a handful of C-ABI calls on
the current HIP stream with every buffer pre-allocated (no autograd graph, no per-step allocation, no host sync:
the loss values stay on the device until somebody asks for them).

Data parallelism: one process per GPU; parameters, gradients and AdamW moments live in ONE flat fp32 buffer
each (layout of ``uvtg_param_offsets``), so gradient averaging is a single bucketed ``all_reduce`` on the flat
buffer (``torch.distributed`` backend "nccl" == RCCL on ROCm).  NCE negatives stay rank-local like the reference.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .dist import allreduce_flat_, allreduce_many_, broadcast_flat_
from .model import Model, SetCriterion, _ptr, _stream

LOSS_KEYS = ("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra")


def flatten_parameters(model: Model) -> torch.Tensor:
    """Re-home every table parameter into one flat fp32 buffer (views keep names/shapes: state_dict is unchanged)."""
    params = model._ordered_params()
    dev = params[0].device
    dims = model._dims(1, 4, 4, model.vid_dim, model.txt_dim, False)
    offs = model._offsets(dims)
    flat = torch.zeros(offs[-1], device=dev)
    for i, p in enumerate(params):
        flat[offs[i]: offs[i] + p.numel()].copy_(p.data.reshape(-1))
        p.data = flat[offs[i]: offs[i] + p.numel()].view(p.shape)
    model._flat = flat
    return flat


class TrainStep:
    def __init__(self, model: Model, criterion: SetCriterion, lr=1e-4, weight_decay=1e-4, grad_clip=0.1,
                 betas=(0.9, 0.999), eps=1e-8, process_group=None, bucket_mb=64, overlap_comm=True, packed="auto", loss_only=True,
                 grad_comm_dtype="fp32", comm_cus=None, time_comm=False):
        if model.precision == "fp32x3":
            raise RuntimeError("training uses the bf16 arithmetic (precision='auto' or 'bf16'); 'fp32x3' is inference-only")
        if "saliency_cls" in criterion.losses:
            raise RuntimeError("the TAL branch ('saliency_cls' + src_cls) runs on the drop-in autograd path (model(..., src_cls=...) + criterion); "
                               "TrainStep implements the dense spans / labels / saliency step")
        self.lib = _lib.load()
        self.model, self.crit = model, criterion
        self.flat = getattr(model, "_flat", None)
        if self.flat is None:
            self.flat = flatten_parameters(model)
        dev = self.flat.device
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.grads = torch.zeros_like(self.flat)
        self.scratch = torch.zeros(1024, device=dev)          # UVTG_ADAMW_SCRATCH_FLOATS (include/uvtg.h)
        self.lr, self.wd, self.clip, self.betas, self.eps = lr, weight_decay, grad_clip, betas, eps
        self.t = 0
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
            broadcast_flat_(self.flat, 0, process_group)
        self.bucket = int(bucket_mb * (1 << 20) // 4)
        # Overlap of the gradient exchange with backward: uvtg_backward records an event when the conv-head gradients and then
        # each encoder layer's gradients are final; every such range is all-reduced on a side stream while the remaining
        # backward kernels keep running (the reference gets the same effect from DDP's bucketed autograd hooks).
        # overlap_comm="force" runs the bucketed side-stream exchange even at world size 1 (single-GPU test of the plumbing)
        self.overlap = bool(overlap_comm) and (self.world > 1 or overlap_comm == "force") and dev.type == "cuda"
        self._events = self._ev_arr = self._comm_stream = None
        self._groups = None
        # gradient buckets on the wire ("bf16 or fp32"): "bf16" halves the bytes every xGMI link carries (174 -> 87 MB per rank
        # and step); every range is rounded to bf16, SUM-reduced in bf16 and widened back (what DDP's bf16_compress_hook does)
        if grad_comm_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_comm_dtype must be 'fp32' or 'bf16'")
        self.grad_comm_dtype = grad_comm_dtype
        self._grads_b = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=dev) if grad_comm_dtype == "bf16" else None
        # CUs kept out of the persistent GEMM grids while a gradient exchange can be in flight (uvtg_set_reserved_cus): RCCL's kernels
        # then never wait for a GEMM workgroup to retire.  Default: UVTG_COMM_CUS, else 0 (the communication kernels co-reside with the
        # GEMM workgroups -- 13 KB of LDS and half the registers of a CU stay free beside a 320-row NT tile)
        if comm_cus is None:
            comm_cus = int(os.environ.get("UVTG_COMM_CUS", "0"))
        self.comm_cus = int(comm_cus) if (self.world > 1 or overlap_comm == "force") else 0
        if dev.type == "cuda":
            self.gemm_cus = self.lib.uvtg_set_reserved_cus(self.comm_cus)
        # time_comm: event pair (end of uvtg_backward on the compute stream, all ranges reduced) -> exposed_comm_ms(): the part of the
        # exchange that backward did NOT hide
        self.time_comm = bool(time_comm)
        self._comm_ev = None
        self._comm_ms = []
        wd = criterion.weight_dict
        self.go = torch.tensor([wd.get(k, 0.0) for k in LOSS_KEYS], dtype=torch.float32, device=dev)
        self.which = (1 if "spans" in criterion.losses else 0) | (2 if "labels" in criterion.losses else 0) | \
                     (4 if "saliency" in criterion.losses else 0)
        # packed (ragged) encoder stream (include/uvtg.h, lens_host): "auto" = when the batch carries the host-side lengths the
        # collate already knows (inputs["_lens_host"] = (lens_v, lens_t)); True = always (lengths read back from the masks: one
        # device->host sync per step); False = padded execution.  Every packed variant gives exactly the padded execution's losses and
        # parameter gradients: under input dropout each padded clip has its own mask, so the engine keeps the valid clips plus the three
        # padded clips the conv heads can see from a valid position (no loss reads anything further out) and the valid text tokens
        self.packed = packed
        # loss_only: this step hands out nothing but losses / gradients, so the packed stream may drop the padded clips no loss can see
        # (include/uvtg.h, dims.loss_only); False keeps every clip row (outputs at padded positions equal the reference's too)
        self.loss_only = bool(loss_only)
        self._lens_arr = None
        self._names = None
        self._shape = None
        self.params = model._ordered_params()
        self.ptrs = model._param_ptrs(self.params)

    def _alloc(self, B, Lv, Lt, dims):
        dev, d = self.flat.device, self.model.hidden_dim
        S = Lv + Lt
        self.ws = torch.empty(self.lib.uvtg_workspace_bytes(C.byref(dims)), dtype=torch.uint8, device=dev)
        self.wcache = torch.empty(self.lib.uvtg_wcache_bytes(C.byref(dims)), dtype=torch.uint8, device=dev)
        self.x0 = torch.empty(B, S, d, device=dev)
        self.pred_logits = torch.empty(B, Lv, 1, device=dev)
        self.pred_spans = torch.empty(B, Lv, 2, device=dev)
        self.txt_mem = torch.empty(B, 1, d, device=dev)
        self.sal = torch.empty(B, Lv, device=dev)
        self.loss_ws = torch.empty(self.lib.uvtg_loss_ws_floats(B, Lv, d), device=dev)
        self.losses = torch.zeros(8, device=dev)
        self.g_logits = torch.empty(B, Lv, device=dev)
        self.g_spans = torch.empty(B, Lv, 2, device=dev)
        self.g_cos = torch.empty(B, Lv, device=dev)
        self.g_vrow = torch.empty(B, d, device=dev)
        self.g_txt = torch.empty(B, d, device=dev)
        self._shape = (B, Lv, Lt)

    def step(self, inputs, targets, optimize=True):
        """One training step; returns the device tensor [loss_b, loss_g, loss_f, loss_s_inter, loss_s_intra] (no sync)."""
        lib, model = self.lib, self.model
        src_txt, src_txt_mask = inputs["src_txt"], inputs["src_txt_mask"]
        src_vid, src_vid_mask = inputs["src_vid"], inputs["src_vid_mask"]
        B, Lv, Dv = src_vid.shape
        Lt, Dt = src_txt.shape[1], src_txt.shape[2]
        dims = model._dims(B, Lv, Lt, Dv, Dt, True)
        dims.loss_only = int(self.loss_only)
        # (re)allocate on a new shape AND whenever the library's own layout for these dims outgrows the buffers: `Model.proj_precise`,
        # `precision`, `n_input_proj`, `use_txt_pos` all change the workspace / operand-cache layout at an unchanged (B, L_v, L_t) -- the split
        # operands take 4 B per element where bf16 takes 2 (flipping proj_precise between steps wrote out of bounds)
        if (self._shape != (B, Lv, Lt) or lib.uvtg_workspace_bytes(C.byref(dims)) > self.ws.numel()
                or lib.uvtg_wcache_bytes(C.byref(dims)) > self.wcache.numel()):
            self._alloc(B, Lv, Lt, dims)
        st = _stream()
        d = model.hidden_dim
        S = Lv + Lt
        chk = _lib.check
        lens = self._host_lens(inputs, B)
        chk(lib.uvtg_prepare_weights(C.byref(dims), self.ptrs, _ptr(self.wcache), st), "uvtg_prepare_weights")
        chk(lib.uvtg_forward(C.byref(dims), self.ptrs, _ptr(self.wcache), _ptr(src_txt), _ptr(src_txt_mask), _ptr(src_vid),
                             _ptr(src_vid_mask), _ptr(model._dim_t(src_vid.device)), _ptr(self.x0), _ptr(self.pred_logits),
                             _ptr(self.pred_spans), _ptr(self.txt_mem), _ptr(self.sal), None, _ptr(self.ws), st, lens), "uvtg_forward")
        tg = targets
        sal = tg.get("saliency_scores")
        pos = tg.get("_pos_idx")
        if pos is None and "saliency_pos_labels" in tg:
            pos = tg["saliency_pos_labels"][:, 0].long().contiguous()
        crit_args = (B, Lv, d, self.which, float(self.crit.eos_coef), _ptr(self.pred_logits), _ptr(self.pred_spans),
                     _ptr(self.x0), S * d, d, _ptr(self.txt_mem), _ptr(tg["timestamp"]), _ptr(tg["timestamp_mask"]),
                     _ptr(tg["timestamp_window"]), _ptr(tg["span_labels_nn"]), _ptr(sal), _ptr(pos), _ptr(self.loss_ws),
                     _ptr(self.losses))
        # cosine / norms of the saliency branch: the forward's saliency pass left them in the workspace (no second pass over vid_mem_proj)
        stats = (C.c_void_p(), C.c_void_p(), C.c_void_p())
        chk(lib.uvtg_forward_saliency_stats(C.byref(dims), _ptr(self.ws), C.byref(stats[0]), C.byref(stats[1]), C.byref(stats[2])),
            "uvtg_forward_saliency_stats")
        chk(lib.uvtg_criterion_fwd(*crit_args, *stats, st), "uvtg_criterion_fwd")
        chk(lib.uvtg_criterion_bwd(*crit_args, _ptr(self.go), _ptr(self.g_logits), _ptr(self.g_spans), None,
                                   _ptr(self.g_txt), _ptr(self.g_cos), _ptr(self.g_vrow), *stats, st), "uvtg_criterion_bwd")
        chk(lib.uvtg_backward(C.byref(dims), self.ptrs, _ptr(self.wcache), _ptr(src_txt), _ptr(src_txt_mask), _ptr(src_vid),
                              _ptr(src_vid_mask), _ptr(self.x0), _ptr(self.pred_logits), _ptr(self.pred_spans), _ptr(self.txt_mem),
                              _ptr(self.g_logits), _ptr(self.g_spans), _ptr(self.g_cos), _ptr(self.g_txt), None, 0, 0,
                              _ptr(self.g_vrow), _ptr(pos), _ptr(self.grads), _ptr(self.ws), st,
                              *self._event_args(dims), lens), "uvtg_backward")
        if self.world > 1 or self.overlap:
            if self.time_comm:
                # one event pair per step, allocated as the steps come: nothing is dropped and nothing synchronises inside a timed
                # loop (round 3 drained an 8-slot ring with a device sync every 8 steps and threw the drained samples away)
                if self._comm_ev is None:
                    self._comm_ev = []
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self._comm_ev.append((e0, e1))
                e0.record()
            self._exchange_gradients(dims)
            if self.time_comm:
                e1.record()
        if optimize:
            self.t += 1
            if self.world == 1 and not self.overlap and not os.environ.get("UVTG_PRENORM_OFF"):
                # single rank: uvtg_backward accumulated the squared gradient norm while writing the gradients
                gn2 = lib.uvtg_backward_gradnorm2(C.byref(dims), _ptr(self.ws))
                chk(lib.uvtg_adamw_clip_step_prenorm(_ptr(self.flat), _ptr(self.grads), _ptr(self.m), _ptr(self.v), self.flat.numel(),
                                                     self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, float(self.clip),
                                                     1.0, gn2, st), "uvtg_adamw_clip_step_prenorm")
            else:
                chk(lib.uvtg_adamw_clip_step(_ptr(self.flat), _ptr(self.grads), _ptr(self.m), _ptr(self.v), self.flat.numel(),
                                             self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, float(self.clip),
                                             1.0 / self.world, _ptr(self.scratch), st), "uvtg_adamw_clip_step")
            model.invalidate_operand_cache()      # the raw kernel rewrote the parameters: model(...) must rebuild its bf16 operands
        return self.losses[:5].clone()            # (the internal buffer is overwritten by the next step)

    # ---- optimizer checkpointing: torch.optim.AdamW state_dict layout, indexed like the reference's optimizer --------------
    def _param_index(self):
        """table parameter -> index in the reference optimizer's single param group
        ([p for n, p in model.named_parameters() if p.requires_grad], main/config.py:349)."""
        idx = {id(p): i for i, (n, p) in enumerate((n, p) for n, p in self.model.named_parameters() if p.requires_grad)}
        return [idx[id(p)] for p in self.params], len(idx)

    def state_dict(self):
        """What ``optimizer.state_dict()`` holds in the reference's checkpoints (main/train_vlp_ddp.py:157-195): loadable by
        ``torch.optim.AdamW.load_state_dict`` and by ``TrainStep.load_state_dict``."""
        index, n = self._param_index()
        offs = self.model._offsets(self.model._dims(1, 4, 4, self.model.vid_dim, self.model.txt_dim, False))
        state = {}
        if self.t > 0:
            for i, p in enumerate(self.params):
                sl = slice(offs[i], offs[i] + p.numel())
                state[index[i]] = {"step": torch.tensor(float(self.t)), "exp_avg": self.m[sl].view_as(p).clone(),
                                   "exp_avg_sq": self.v[sl].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(n))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        index, _ = self._param_index()
        offs = self.model._offsets(self.model._dims(1, 4, 4, self.model.vid_dim, self.model.txt_dim, False))
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.wd = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for i, p in enumerate(self.params):
            ent = sd["state"].get(index[i])
            if ent is None:
                continue
            sl = slice(offs[i], offs[i] + p.numel())
            self.m[sl].copy_(ent["exp_avg"].reshape(-1))
            self.v[sl].copy_(ent["exp_avg_sq"].reshape(-1))
            steps.add(int(float(ent["step"])))
        if len(steps) > 1:
            raise ValueError("TrainStep keeps ONE step count for all parameters (the reference's optimizer steps them together)")
        self.t = steps.pop() if steps else 0


    def _host_lens(self, inputs, B):
        """ctypes int[2B] (clips then text tokens per sample) for the packed encoder stream, or None for padded execution."""
        if self.packed is False:
            return None
        lens = inputs.get("_lens_host")
        if lens is None:
            if self.packed != True:      # noqa: E712  ("auto": only with caller-provided lengths)
                return None
            lens = torch.stack([inputs["src_vid_mask"].sum(1), inputs["src_txt_mask"].sum(1)]).to(torch.int32).cpu()   # sync
        lv, lt = lens[0], lens[1]
        vals = [int(x) for x in lv] + [int(x) for x in lt]
        if len(vals) != 2 * B:
            raise ValueError("_lens_host must hold B clip counts and B token counts")
        Lv, Lt = inputs["src_vid"].shape[1], inputs["src_txt"].shape[1]
        if self.packed == "auto" and all(v == Lv for v in vals[:B]) and all(v == Lt for v in vals[B:]):
            return None                  # nothing is padded: the packed stream would be the same rows behind gather tables
        self._lens_arr = (C.c_int * (2 * B))(*vals)     # read synchronously by the engine (row count only; tables come from the masks)
        return self._lens_arr

    # ---- data-parallel gradient exchange -------------------------------------------------------------------------
    def bucket_ranges(self, dims):
        """Element ranges of the flat gradient buffer in the order they become final during uvtg_backward:
        [conv heads] + [layer E-1, ..., layer 0] (one event each) and the rest (token-type, input projections, pool)."""
        offs = self.model._offsets(dims)
        E = self.model.enc_layers
        ranged = [(offs[12 * E + 1], offs[12 * E + 13])] + [(offs[12 * l], offs[12 * (l + 1)]) for l in range(E - 1, -1, -1)]
        rest = [(offs[12 * E], offs[12 * E + 1]), (offs[12 * E + 13], offs[-1])]
        return ranged, rest

    def _event_args(self, dims):
        if not self.overlap:
            return None, 0
        if self._events is None:
            n = self.model.enc_layers + 1
            self._events = [torch.cuda.Event() for _ in range(n)]
            for ev in self._events:
                ev.record()                                  # materialise the hipEvent_t
            self._ev_arr = (C.c_void_p * n)(*[ev.cuda_event for ev in self._events])
            self._comm_stream = torch.cuda.Stream()
        return self._ev_arr, len(self._events)

    def exposed_comm_ms(self):
        """Per-step exposed communication times (ms) recorded since the last call (time_comm=True): from the end of uvtg_backward on the
        compute stream to the moment every gradient range is reduced.  Synchronises."""
        if self._comm_ev:
            torch.cuda.synchronize()
            self._comm_ms += [a.elapsed_time(b) for a, b in self._comm_ev]
            self._comm_ev = []
        out, self._comm_ms = self._comm_ms, []
        return out

    def flat_checksum(self):
        """(sum, sum of squares, first / middle / last element) of the flat parameter buffer in float64 -- data-parallel replicas must agree
        on it bit for bit after any number of steps (bench.py --gpus N prints the comparison).  Synchronises."""
        f = self.flat.double()
        n = f.numel()
        return [float(f.sum()), float((f * f).sum()), float(f[0]), float(f[n // 2]), float(f[n - 1])]

    def _reduce_range(self, lo, hi):
        """SUM all-reduce of grads[lo:hi] on the CURRENT stream, in the configured wire dtype."""
        self._reduce_ranges([(lo, hi)])

    def _reduce_ranges(self, ranges):
        """SUM all-reduce of several ranges of the flat gradient buffer on the CURRENT stream as ONE coalesced collective (round 6): ranges that
        touch are merged first (the encoder layers of one readiness group are one contiguous block of the buffer)."""
        merged = []
        for lo, hi in sorted((lo, hi) for lo, hi in ranges if hi > lo):
            if merged and merged[-1][1] == lo:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        if not merged:
            return
        if os.environ.get("UVTG_COMM_COALESCE_OFF"):
            merged = [[lo, hi] for lo, hi in ranges if hi > lo]
            if self._grads_b is None:
                for lo, hi in merged:
                    allreduce_flat_(self.grads[lo:hi], self.bucket, self.pg)
                return
        if self._grads_b is None:
            allreduce_many_([self.grads[lo:hi] for lo, hi in merged], self.bucket, self.pg)
            return
        st = _stream()
        for lo, hi in merged:
            _lib.check(self.lib.uvtg_cast_bf16(_ptr(self.grads[lo:hi]), _ptr(self._grads_b[lo:hi]), hi - lo, st), "uvtg_cast_bf16")
        allreduce_many_([self._grads_b[lo:hi] for lo, hi in merged], 2 * self.bucket, self.pg)        # (same bytes per collective as the fp32 buckets)
        for lo, hi in merged:
            _lib.check(self.lib.uvtg_cast_f32(_ptr(self._grads_b[lo:hi]), _ptr(self.grads[lo:hi]), hi - lo, st), "uvtg_cast_f32")

    def _event_groups(self):
        """Last event index of every readiness group (uvtg_backward_event_groups): the library records the events of a group together."""
        if self._groups is None and os.environ.get("UVTG_COMM_COALESCE_OFF"):      # (A/B: one wait + one collective per range, rounds 2-5)
            self._groups = list(range(self.model.enc_layers + 1))
        if self._groups is None:
            E = self.model.enc_layers
            buf = (C.c_int * (E + 1))()
            n = self.lib.uvtg_backward_event_groups(E, buf)
            if n <= 0:
                raise RuntimeError(f"uvtg_backward_event_groups failed ({n})")
            self._groups = [int(buf[i]) for i in range(n)]
        return self._groups

    def _exchange_gradients(self, dims):
        if not self.overlap:
            self._reduce_range(0, self.grads.numel())
            return
        ranged, rest = self.bucket_ranges(dims)
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._comm_stream):
            first = 0
            for last in self._event_groups():                # one wait + ONE coalesced collective per readiness group
                self._comm_stream.wait_event(self._events[last])      # the group's ranges are final on the compute stream
                self._reduce_ranges(ranged[first: last + 1])
                first = last + 1
            self._comm_stream.wait_stream(main)              # end of backward: everything else is final
            self._reduce_ranges(rest)
        main.wait_stream(self._comm_stream)                  # the optimizer step needs every reduced range
