"""HIP-graph replay of the inference call (model forward under no_grad + device post-processing).

The reference's evaluation loop (main/inference_mr.py:88-193) issues one model call + a Python post-processing loop per batch of
`--eval_bsz 32` samples.  On MI355X the precise forward of such a batch is ~75 kernel dispatches of a few microseconds each: at
batch 1 .. 32 the call is bound by launch latency and the Python boundary, not by the kernels.  libuvtg.so never allocates and never
synchronises (include/uvtg.h), so the whole call can be captured ONCE per input shape into a HIP graph and replayed:

    run = GraphedInference(model, clip_length=2.0)            # model: univtg_amd.model.Model in eval mode
    out = run(src_txt, src_txt_mask, src_vid, src_vid_mask, timestamp, timestamp_mask, durations)
    out["pred_logits"], out["windows"], out["order"], out["keep"], out["n_keep"], out["saliency"]

The returned tensors are the graph's static output buffers: consume (or clone) them before the next call of the same shape
(`clone_outputs=True` hands out copies instead).  There is no fallback: shapes are captured on first use (two eager warm-up calls on a side
stream, then the capture).  Every captured shape keeps its own workspace (incl. the 32 MB split-K slab) inside the graph's memory pool, and the
reference's evaluation loop pads each batch to its own maximum length: the cache is an LRU of `max_graphs` shapes (default 8; the least
recently used graph and its pool are dropped -- unless the caller still holds the static output tensors handed out for that shape
(`clone_outputs=False`): those keep the evicted pool's blocks alive until they are released; `stats` counts hits, captures and evictions,
and a loader that cycles through more than `max_graphs` distinct shapes re-captures on every miss (2 eager warm-up calls + a device
synchronisation + the capture): size `max_graphs` for the shapes of one evaluation pass or call the model eagerly).  (Padding the lengths up to a few bucket sizes would let shapes share graphs, but it is NOT exact:
the k = 3 conv heads of a maximum-length sample see encoder outputs of padded clips where the unpadded call sees the zero frame -- measured
0.23 on `pred_logits` at the last valid clips -- so it is not offered.)
"""
from __future__ import annotations

import torch

from . import ops
from .model import Model

__all__ = ["GraphedInference"]


class GraphedInference:
    def __init__(self, model: Model, clip_length: float = 2.0, eval_mode: str = "add", nms_thd: float = 0.7, max_before: int = 1000,
                 max_after: int = 10, max_graphs: int = 8, clone_outputs: bool = False):
        if model.training:
            raise RuntimeError("GraphedInference captures the inference call: put the model in eval() mode first")
        if model.packed:
            raise RuntimeError("the packed stream reads the mask sums back to the host (a sync): capture needs Model(packed=False)")
        self.model = model
        self.post = dict(clip_length=clip_length, eval_mode=eval_mode, nms_thd=nms_thd, max_before=max_before, max_after=max_after)
        self._graphs = {}                                   # insertion-ordered: least recently used first
        self.max_graphs = max(1, int(max_graphs))
        self.clone_outputs = bool(clone_outputs)
        self.stats = dict(hits=0, captures=0, evictions=0, recaptures_after_parameter_update=0)

    def _call(self, st):
        with torch.no_grad():
            out = self.model(st["src_txt"], st["src_txt_mask"], st["src_vid"], st["src_vid_mask"])
            win, order, keep, nk, sal = ops.postprocess_mr(out["pred_logits"], out["pred_spans"], out["saliency_scores"], st["timestamp"],
                                                           st["timestamp_mask"], st["durations"], **self.post)
        return dict(pred_logits=out["pred_logits"], pred_spans=out["pred_spans"], saliency_scores=out["saliency_scores"],
                    windows=win, order=order, keep=keep, n_keep=nk, saliency=sal)

    def _capture(self, key, tensors):
        static = {k: v.clone() for k, v in tensors.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up: operand cache, one-time function attributes, lazily built tables
            for _ in range(2):
                self._call(static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()                      # (a hipGraph on ROCm)
        with torch.cuda.graph(graph):
            outs = self._call(static)
        self._graphs.pop(key, None)
        while len(self._graphs) >= self.max_graphs:         # LRU eviction: drop the oldest graph (and with it its memory pool)
            self._graphs.pop(next(iter(self._graphs)))
            self.stats["evictions"] += 1
        self.stats["captures"] += 1
        self._graphs[key] = (graph, static, outs, self.model._param_epoch, tuple(p._version for p in self.model._ordered_params()))
        return self._graphs[key]

    def __call__(self, src_txt, src_txt_mask, src_vid, src_vid_mask, timestamp, timestamp_mask, durations):
        tensors = dict(src_txt=src_txt, src_txt_mask=src_txt_mask, src_vid=src_vid, src_vid_mask=src_vid_mask, timestamp=timestamp,
                       timestamp_mask=timestamp_mask, durations=durations)
        for k, v in tensors.items():
            if not v.is_cuda:
                raise RuntimeError(f"{k}: inputs must live on the ROCm device (no CPU fallback)")
            if v.dtype != torch.float32:
                tensors[k] = v.float()
        key = tuple((k, tuple(v.shape)) for k, v in tensors.items())
        ent = self._graphs.get(key)
        sig = (self.model._param_epoch, tuple(p._version for p in self.model._ordered_params()))
        if ent is not None and (ent[3], ent[4]) != sig:     # parameters changed: the captured call holds the old bf16 / split operands
            ent = None
            self.stats["recaptures_after_parameter_update"] += 1
        if ent is None:
            ent = self._capture(key, tensors)
        else:
            self.stats["hits"] += 1
        self._graphs[key] = self._graphs.pop(key)            # most recently used
        graph, static, outs = ent[:3]
        for k, v in tensors.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return {k: v.clone() for k, v in outs.items()} if self.clone_outputs else outs
