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

Staging: every key owns a RING of pinned host buffers; a buffer is reused only after the event recorded behind its last H2D
copy has fired (the host runs ahead of a training loop that never synchronises -- a single buffer per key would be refilled
while its previous copy is still queued behind the running step).

``pack_batch_host`` / ``upload_packed_batch`` (round 6) split that call at the PCIe boundary: the packing -- one memcpy of the whole batch --
runs as the ``collate_fn`` of the loader's worker processes into pinned memory, the training process only enqueues the copies.

``DevicePrefetcher(batches, device)`` wraps any iterable of sample lists (a ``DataLoader`` with ``collate_fn=lambda b: b``) or of
``PackedHostBatch`` es (``collate_fn=pack_batch_host``) and
uploads batch n + 1 on a side stream while the caller computes on batch n: the H2D copies and the padding kernels overlap the
step instead of preceding it (the reference's loop does the copy synchronously inside the step, main/train_vlp_ddp.py:52).
"""
from __future__ import annotations

import collections
import time

import numpy as np
import torch

from . import _lib
from .model import _ptr, _stream

_PADDED_KEYS = ("query_feat", "video_feat", "timestamp", "timestamp_window", "span_labels_nn", "saliency_scores", "weight_ablation")


class _Staging:
    """Per key: a ring of grow-only pinned host buffers, each guarded by the event recorded after its last H2D copy."""

    def __init__(self, depth=3):
        self.depth = depth
        self.ring = {}           # key -> list of [buffer, event-or-None]
        self.next = {}
        self.waits = 0           # how often the host had to wait for a slot (ran more than `depth` uploads ahead)

    def acquire(self, key, numel, dtype):
        ring = self.ring.setdefault(key, [[None, None] for _ in range(self.depth)])
        i = self.next.get(key, 0)
        self.next[key] = (i + 1) % self.depth
        slot = ring[i]
        if slot[1] is not None and not slot[1].query():
            self.waits += 1
            slot[1].synchronize()                     # the copy out of this buffer is still in flight
        if slot[0] is None or slot[0].numel() < numel or slot[0].dtype != dtype:
            slot[0] = torch.empty(max(numel, 1), dtype=dtype).pin_memory()
        return slot

    @staticmethod
    def release(slot, device=None):
        """Call right after enqueueing the H2D copy that reads the slot's buffer: the event is recorded on the TARGET device's current
        stream -- the stream the copy runs on (with cuda:0 current and an upload to cuda:1 the default record() would not cover it)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if slot[1] is None or getattr(slot[1], "_uvtg_dev", None) != dev.index:
            with torch.cuda.device(dev):
                slot[1] = torch.cuda.Event()
            slot[1]._uvtg_dev = dev.index
        slot[1].record(torch.cuda.current_stream(dev))


_staging = _Staging()


def _upload(key, host_tensor_fn, numel, dtype, device):
    """pinned slot <- host_tensor_fn(out=view), async H2D on the current stream, slot guarded by an event."""
    slot = _staging.acquire(key, numel, dtype)
    view = slot[0][:numel]
    host_tensor_fn(view)
    dev = view.to(device, non_blocking=True)
    _staging.release(slot, device)
    return dev


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
    packed = _upload(key, lambda out: torch.cat([s.reshape(s.shape[0], -1).to(wire_dtype) for s in seqs], 0, out=out.view(total, D)),
                     total * D, wire_dtype, device)
    offs = [0]
    for n in lengths:
        offs.append(offs[-1] + n)
    offsets = _upload(key + "/offsets", lambda out: out.copy_(torch.tensor(offs, dtype=torch.int32)), B + 1, torch.int32, device)
    out = torch.empty((B, Lmax) + extra, dtype=torch.float32, device=device)
    mask = torch.empty(B, Lmax, dtype=torch.float32, device=device)
    _lib.check(lib.uvtg_ragged_to_padded(_ptr(packed), int(wire_dtype == torch.bfloat16), _ptr(offsets), B, Lmax, D, _ptr(out), _ptr(mask),
                                         _stream()), "uvtg_ragged_to_padded")
    return out, mask, lengths


def collate_upload_mr(batch, device, feature_dtype=torch.float32):
    """(batch_meta, model_inputs, targets): the reference's collate + device upload for a list of dataset samples.
    All device work is enqueued on the CURRENT stream; nothing waits for it."""
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
            t = torch.LongTensor(vals)
            data[k] = _upload(k, lambda out, t=t: out.view(t.shape).copy_(t), t.numel(), torch.int64, device).view(t.shape)
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


class PackedHostBatch:
    """One collated batch ON THE HOST in the wire format (round 6): per padded key ONE packed ``[sum(len), D]`` block in pinned memory plus
    the lengths, the label keys as pinned tensors.  What a DataLoader worker (``collate_fn=pack_batch_host``, ``pin_memory=True``) hands to the
    main process: the 216 MB memcpy of the collate then runs in the workers, and the training process only enqueues the H2D copies
    (``upload_packed_batch``) -- at config 2 one process packs ~10 GB/s, i.e. 20 ms per batch against an 9 ms step."""

    def __init__(self, meta, padded, labels, span_labels):
        self.meta, self.padded, self.labels, self.span_labels = meta, padded, labels, span_labels

    def pin_memory(self):                            # (torch's DataLoader calls this on custom batch types with pin_memory=True)
        for k, (blk, lengths, extra, offs) in self.padded.items():
            self.padded[k] = (blk if blk.is_pinned() else blk.pin_memory(), lengths, extra, offs if offs.is_pinned() else offs.pin_memory())
        self.labels = {k: (v if v.is_pinned() else v.pin_memory()) for k, v in self.labels.items()}
        return self


def pack_batch_host(batch, feature_dtype=torch.float32, pin=True):
    """The host half of ``collate_upload_mr`` as a ``collate_fn``: list of dataset samples -> ``PackedHostBatch`` (no device work)."""
    meta = [e["meta"] for e in batch]
    padded, labels, span_labels = {}, {}, None
    for k in batch[0]["model_inputs"].keys():
        vals = [e["model_inputs"][k] for e in batch]
        if k == "span_labels":
            span_labels = [torch.as_tensor(v, dtype=torch.float32) for v in vals]
        elif k in ("saliency_pos_labels", "saliency_neg_labels"):
            labels[k] = torch.LongTensor(vals)
        else:
            wire = feature_dtype if k in ("query_feat", "video_feat") else torch.float32
            seqs = [torch.as_tensor(v) for v in vals]
            lengths = [int(q.shape[0]) for q in seqs]
            extra = tuple(seqs[0].shape[1:])
            blk = torch.cat([q.reshape(q.shape[0], -1).to(wire) for q in seqs], 0)
            offs = torch.tensor([0] + list(np.cumsum(lengths)), dtype=torch.int32)
            padded[k] = (blk, lengths, extra, offs)
    pb = PackedHostBatch(meta, padded, labels, span_labels)
    return pb.pin_memory() if pin else pb


def upload_packed_batch(pb, device):
    """The device half: async H2D copy of every packed block (straight out of the batch's own pinned memory: the caller keeps ``pb`` alive
    until the copies are done -- DevicePrefetcher does) + ``uvtg_ragged_to_padded``; returns what ``collate_upload_mr`` returns."""
    lib = _lib.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("univtg_amd.pipeline uploads to an MI355X: device must be a ROCm device (no CPU fallback)")
    data, lens = {}, {}
    for k, (blk, lengths, extra, offs) in pb.padded.items():
        B, Lmax, D = len(lengths), max(lengths), int(blk.shape[1])
        packed = blk.to(device, non_blocking=True)
        offsets = offs.to(device, non_blocking=True)
        out = torch.empty((B, Lmax) + extra, dtype=torch.float32, device=device)
        mask = torch.empty(B, Lmax, dtype=torch.float32, device=device)
        _lib.check(lib.uvtg_ragged_to_padded(_ptr(packed), int(blk.dtype == torch.bfloat16), _ptr(offsets), B, Lmax, D, _ptr(out), _ptr(mask),
                                             _stream()), "uvtg_ragged_to_padded")
        data[k], lens[k] = (out, mask), lengths
    model_inputs = dict(src_txt=data["query_feat"][0], src_txt_mask=data["query_feat"][1],
                        src_vid=data["video_feat"][0], src_vid_mask=data["video_feat"][1],
                        _lens_host=(lens["video_feat"], lens["query_feat"]))
    targets = dict(timestamp=data["timestamp"][0], timestamp_mask=data["timestamp"][1],
                   timestamp_window=data["timestamp_window"][0], span_labels_nn=data["span_labels_nn"][0])
    if "saliency_scores" in data:
        targets["saliency_scores"] = data["saliency_scores"][0]
    if pb.span_labels is not None:
        targets["span_labels"] = [dict(spans=v.to(device, non_blocking=True)) for v in pb.span_labels]
    for k, v in pb.labels.items():
        targets[k] = v.to(device, non_blocking=True)
    if "weight_ablation" in data:
        targets["weight_ablation"] = data["weight_ablation"][0]
    return pb.meta, model_inputs, targets


def _device_tensors(obj):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _device_tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _device_tensors(v)


class DevicePrefetcher:
    """Iterate ``(meta, model_inputs, targets)`` over an iterable of sample lists, keeping ``depth`` uploads in flight on a side
    stream.  The consumer's stream waits for the upload's event only (no host synchronisation); uploaded tensors are tied to the
    consumer's stream with ``record_stream`` so that the caching allocator does not recycle them under a running step.

    ``stats`` after (or during) iteration: ``host_collate_s`` (host time spent packing + enqueueing), ``batches`` and
    ``upload_ms`` -- device time of the uploads measured by events on the side stream -- all of which is hidden behind the
    consumer's compute as long as the consumer's step is longer than the upload."""

    def __init__(self, batches, device, feature_dtype=torch.float32, depth=2, timing=False):
        self.batches, self.device, self.feature_dtype, self.depth, self.timing = batches, torch.device(device), feature_dtype, depth, timing
        self.stream = torch.cuda.Stream(device=self.device)
        self.stats = dict(host_collate_s=0.0, batches=0, upload_ms=0.0)
        self._timers = []
        self._alive = collections.deque()

    def _enqueue(self, samples):
        t0 = time.perf_counter()
        with torch.cuda.stream(self.stream):
            if self.timing:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if isinstance(samples, PackedHostBatch):       # collated (and pinned) by the loader's workers: only the copies are enqueued here
                item = upload_packed_batch(samples, self.device)
                self._alive.append(samples)                 # its pinned blocks must outlive the copies (dropped `depth + 2` batches later)
                while len(self._alive) > self.depth + 2:
                    self._alive.popleft()
            else:
                item = collate_upload_mr(samples, self.device, self.feature_dtype)
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record()
            if self.timing:
                self._timers.append((e0, ev))
        self.stats["host_collate_s"] += time.perf_counter() - t0
        self.stats["batches"] += 1
        return item, ev

    def __iter__(self):
        q = collections.deque()
        it = iter(self.batches)
        done = False
        while True:
            while not done and len(q) < self.depth:
                try:
                    q.append(self._enqueue(next(it)))
                except StopIteration:
                    done = True
            if not q:
                break
            item, ev = q.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in _device_tensors(item):
                t.record_stream(cur)
            yield item
        if self.timing:
            torch.cuda.synchronize(self.device)
            self.stats["upload_ms"] = sum(a.elapsed_time(b) for a, b in self._timers)
            self._timers = []


# ---- feature files (main/dataset.py:325-358 query, :370-390 video) ------------------------------------------
def _l2_normalize(a, eps=1e-5):
    """utils/basic_utils.py:97-99"""
    return a / (np.linalg.norm(a, axis=-1, keepdims=True) + eps)


def read_video_features(paths, normalize=True):
    """The reference's ``_get_video_feat_by_vid`` without the hdf5 cache: one ``.npz`` per feature type (key ``features``, e.g.
    SlowFast + CLIP), each optionally l2-normalised per clip, truncated to the shortest ("some features are slightly longer than the
    others") and concatenated along the feature axis -> fp32 tensor (L_v, sum D)."""
    feats = []
    for path in paths:
        f = np.load(path)["features"].astype(np.float32)
        feats.append(_l2_normalize(f) if normalize else f)
    n = min(len(f) for f in feats)
    return torch.from_numpy(np.concatenate([f[:n] for f in feats], axis=1))


def read_query_features(path, feat_type="last_hidden_state", normalize=True, feat_dim=512):
    """The reference's ``_get_query_feat_by_qid`` without the cache and without the (training-time, unused by the scripts) row drop:
    key ``last_hidden_state`` (L_q, D) or ``pooler_output`` (D,); an unreadable file gives the reference's zeros((10, D)) placeholder."""
    try:
        q = np.load(path)[feat_type].astype(np.float32)
    except Exception:
        q = np.zeros((10, feat_dim), np.float32)
    return torch.from_numpy(_l2_normalize(q) if normalize else q)


def load_feature_cache(store, keys, optional=False):
    """The reference's in-memory feature cache (``DatasetVLP.__init__`` with ``use_cache``, main/dataset.py:113-131): ``store`` is any
    mapping whose items slice to arrays -- an open ``h5py.File`` of ``data/<dset>/h5py/<feat_type>.hdf5`` when the caller has h5py (this
    package does not import it), or a dict of arrays -- and the result holds ``store[str(key)][:]`` for every key.  ``optional`` skips keys
    the store does not have (the reference does that for the text cache and substitutes zeros at read time)."""
    out = {}
    for key in keys:
        try:
            out[key] = np.asarray(store[str(key)][:])
        except Exception:
            if not optional:
                raise
    return out


def read_video_features_cached(caches, vid):
    """``_get_video_feat_by_vid`` on the cache path (main/dataset.py:375-376,382-386): one cache per feature type, entries taken AS STORED
    (the hdf5 files hold the features the way ``data/create_h5py.py`` wrote them: no cast, no normalisation here), truncated to the
    shortest and concatenated along the feature axis."""
    feats = [np.asarray(c[vid]) for c in caches]
    n = min(len(f) for f in feats)
    return torch.from_numpy(np.concatenate([f[:n] for f in feats], axis=1))


def read_query_features_cached(cache, qid, feat_dim=512):
    """``_get_query_feat_by_qid`` on the cache path (main/dataset.py:335-340): the stored array as is, zeros((10, D)) for a missing qid."""
    try:
        q = np.asarray(cache[qid])
    except Exception:
        q = np.zeros((10, feat_dim), np.float32)
    return torch.from_numpy(q)

