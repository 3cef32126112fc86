"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's collate / wire format for one batch
(SURVEY 8f row 2): ``pad_sequences_1d`` (utils/tensor_utils.py:5-53), ``start_end_collate_mr``
(main/dataset.py:1037-1052) and ``prepare_batch_inputs_mr`` (main/dataset.py:1071-1100).
Pinned against tests/golden/collate.npz, which oracle/make_golden.py produced by running the real reference functions."""
from __future__ import annotations

import numpy as np


def pad_sequences_1d(seqs, dtype=np.float32):
    """utils/tensor_utils.py:5-53: zero-pad along the first dim to the batch maximum; mask 1 = valid."""
    seqs = [np.asarray(s) for s in seqs]
    lengths = [len(s) for s in seqs]
    extra = seqs[0].shape[1:]
    out = np.zeros((len(seqs), max(lengths)) + extra, dtype=dtype)
    mask = np.zeros((len(seqs), max(lengths)), dtype=np.float32)
    for i, s in enumerate(seqs):
        out[i, :lengths[i]] = s          # (float64 saliency scores are cast to fp32 here, as torch does)
        mask[i, :lengths[i]] = 1
    return out, mask


def collate_mr(batch):
    """main/dataset.py:1037-1052 + 1071-1100 -> (model_inputs, targets) as numpy."""
    keys = batch[0]["model_inputs"].keys()
    data = {}
    for k in keys:
        vals = [e["model_inputs"][k] for e in batch]
        if k == "span_labels":
            data[k] = [np.asarray(v, dtype=np.float32) for v in vals]
        elif k in ("saliency_pos_labels", "saliency_neg_labels"):
            data[k] = np.asarray(vals, dtype=np.int64)
        else:
            data[k] = pad_sequences_1d(vals)
    model_inputs = dict(src_txt=data["query_feat"][0], src_txt_mask=data["query_feat"][1],
                        src_vid=data["video_feat"][0], src_vid_mask=data["video_feat"][1])
    targets = dict(timestamp=data["timestamp"][0], timestamp_mask=data["timestamp"][1],
                   timestamp_window=data["timestamp_window"][0], span_labels_nn=data["span_labels_nn"][0])
    if "saliency_scores" in data:
        targets["saliency_scores"] = data["saliency_scores"][0]
    if "span_labels" in data:
        targets["span_labels"] = data["span_labels"]
    for k in ("saliency_pos_labels", "saliency_neg_labels"):
        if k in data:
            targets[k] = data[k]
    return model_inputs, targets
