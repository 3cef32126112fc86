"""Input pipeline / wire format for the hot path on MI355X.

``collate_upload_mr(batch, device)`` takes the list of per-sample dicts that the reference's datasets return
(``DatasetVLP/DatasetMR.__getitem__``, main/dataset.py:153-240) and produces what
``prepare_batch_inputs_mr(start_end_collate_mr(batch)[1], device)`` produces (main/dataset.py:1037-1052,1071-1100) --
the padded ``model_inputs`` / ``targets`` tensors on the device -- but

* only the VALID rows cross PCIe: every padded key travels as one packed ``[sum(len), D]`` block from a pinned
  staging buffer (one async H2D copy per key) and is expanded to the zero-padded batch + mask by ``uvtg_ragged_to_padded``
  on the device (the reference pads on the host and ships the zeros: 216 MB fp32 per config-2 step, ~25 % of it padding);
* the per-sample lengths the collate knows anyway are handed on as ``model_inputs["_lens_host"]`` so that the engine can run
  its packed (ragged) encoder stream without a device->host sync;
* ``feature_dtype=torch.bfloat16`` optionally halves the feature bytes on the wire (not bit-exact: the features are rounded).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .model import _ptr, _stream

_PADDED_KEYS = ("query_feat", "video_feat", "timestamp", "timestamp_window", "span_labels_nn", "saliency_scores", "weight_ablation")


class _Staging:
    """Grow-only pinned host buffers, one per key (reused across batches)."""

    def __init__(self):
        self.buf = {}

    def get(self, key, numel, dtype):
        b = self.buf.get(key)
        if b is None or b.numel() < numel or b.dtype != dtype:
            b = torch.empty(max(numel, 1), dtype=dtype).pin_memory()
            self.buf[key] = b
        return b[:numel]


_staging = _Staging()


def _pad_on_device(key, seqs, device, wire_dtype):
    """One key: pack valid rows -> pinned staging -> async H2D -> uvtg_ragged_to_padded.  Returns (padded fp32, mask fp32, lengths)."""
    lib = _lib.load()
    seqs = [torch.as_tensor(s) for s in seqs]
    lengths = [int(s.shape[0]) for s in seqs]
    extra = tuple(seqs[0].shape[1:])
    D = 1
    for e in extra:
        D *= int(e)
    total, B, Lmax = sum(lengths), len(seqs), max(lengths)
    stage = _staging.get(key, total * D, wire_dtype)
    torch.cat([s.reshape(s.shape[0], -1).to(wire_dtype) for s in seqs], 0, out=stage.view(total, D))
    packed = stage.to(device, non_blocking=True)
    offs = [0]
    for n in lengths:
        offs.append(offs[-1] + n)
    offsets = torch.tensor(offs, dtype=torch.int32).to(device, non_blocking=True)
    out = torch.empty((B, Lmax) + extra, dtype=torch.float32, device=device)
    mask = torch.empty(B, Lmax, dtype=torch.float32, device=device)
    _lib.check(lib.uvtg_ragged_to_padded(_ptr(packed), int(wire_dtype == torch.bfloat16), _ptr(offsets), B, Lmax, D, _ptr(out), _ptr(mask),
                                         _stream()), "uvtg_ragged_to_padded")
    return out, mask, lengths


def collate_upload_mr(batch, device, feature_dtype=torch.float32):
    """(batch_meta, model_inputs, targets): the reference's collate + device upload for a list of dataset samples."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("univtg_amd.pipeline uploads to an MI355X: device must be a ROCm device (no CPU fallback)")
    meta = [e["meta"] for e in batch]
    keys = batch[0]["model_inputs"].keys()
    data, lens = {}, {}
    for k in keys:
        vals = [e["model_inputs"][k] for e in batch]
        if k == "span_labels":
            data[k] = [dict(spans=torch.as_tensor(v, dtype=torch.float32).to(device, non_blocking=True)) for v in vals]
        elif k in ("saliency_pos_labels", "saliency_neg_labels"):
            data[k] = torch.LongTensor(vals).to(device, non_blocking=True)
        else:
            wire = feature_dtype if k in ("query_feat", "video_feat") else torch.float32
            data[k] = _pad_on_device(k, vals, device, wire)
            lens[k] = data[k][2]
    model_inputs = dict(src_txt=data["query_feat"][0], src_txt_mask=data["query_feat"][1],
                        src_vid=data["video_feat"][0], src_vid_mask=data["video_feat"][1],
                        _lens_host=(lens["video_feat"], lens["query_feat"]))
    targets = dict(timestamp=data["timestamp"][0], timestamp_mask=data["timestamp"][1],
                   timestamp_window=data["timestamp_window"][0], span_labels_nn=data["span_labels_nn"][0])
    if "saliency_scores" in data:
        targets["saliency_scores"] = data["saliency_scores"][0]
    if "span_labels" in data:
        targets["span_labels"] = data["span_labels"]
    for k in ("saliency_pos_labels", "saliency_neg_labels"):
        if k in data:
            targets[k] = data[k]
    if "weight_ablation" in data:
        targets["weight_ablation"] = data["weight_ablation"][0]
    return meta, model_inputs, targets
