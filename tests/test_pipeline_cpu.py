"""Host half of the input pipeline (CPU-only): `pack_batch_host` as the collate_fn of DataLoader WORKER processes -- the wire format must survive
the trip to the training process (pickling through the worker queue) and equal what the reference's collate pads (golden collate.npz made by the
real start_end_collate_mr, main/dataset.py:1037-1052): per key, the valid rows of every sample back to back + the prefix lengths."""
import functools
import os

import numpy as np
import torch

from test_oracle_golden import _collate_case


class _Samples(torch.utils.data.Dataset):
    def __init__(self, batch):
        self.batch = batch

    def __len__(self):
        return len(self.batch)

    def __getitem__(self, i):
        return self.batch[i]


def test_pack_batch_host_in_loader_workers_matches_the_reference_collate(golden_dir):
    from univtg_amd.pipeline import PackedHostBatch, pack_batch_host
    z, batch = _collate_case(golden_dir)
    for e in batch:
        e["model_inputs"] = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in e["model_inputs"].items()}
    loader = torch.utils.data.DataLoader(_Samples(batch), batch_size=len(batch), shuffle=False, num_workers=1,
                                         collate_fn=functools.partial(pack_batch_host, feature_dtype=torch.float32, pin=False))
    (pb,) = list(loader)
    assert isinstance(pb, PackedHostBatch) and [m["qid"] for m in pb.meta] == list(range(len(batch)))
    for key, ref_key, mask_key in (("video_feat", "in/src_vid", "in/src_vid_mask"), ("query_feat", "in/src_txt", "in/src_txt_mask"),
                                   ("timestamp", "tg/timestamp", "tg/timestamp_mask")):
        blk, lengths, extra, offs = pb.padded[key]
        ref, mask = z[ref_key], z[mask_key]
        assert lengths == [int(x) for x in mask.sum(1)] and offs.tolist() == [0] + list(np.cumsum(lengths))
        want = np.concatenate([ref[b, :lengths[b]].reshape(lengths[b], -1) for b in range(len(lengths))], 0)
        assert np.array_equal(blk.numpy(), want), key
        assert tuple(extra) == tuple(ref.shape[2:])
    assert np.array_equal(pb.labels["saliency_pos_labels"].numpy(), z["tg/saliency_pos_labels"])
    # bf16 on the wire: only the features are rounded
    pb16 = pack_batch_host(batch, feature_dtype=torch.bfloat16, pin=False)
    assert pb16.padded["video_feat"][0].dtype == torch.bfloat16 and pb16.padded["timestamp"][0].dtype == torch.float32
    assert torch.equal(pb16.padded["video_feat"][0].float(), pb.padded["video_feat"][0].to(torch.bfloat16).float())
