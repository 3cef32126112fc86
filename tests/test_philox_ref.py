"""Known-answer vectors (Random123 kat_vectors, philox4x32-10) for the test-side Philox restatement that
tests/test_gpu_model.py::test_train_mode_dropout_replayed_through_oracle uses to regenerate the device's dropout masks."""
import numpy as np

import philox_ref as R


def test_philox4x32_10_known_answers():
    z = R.philox4(0, np.array([0], dtype=np.uint64), 0, 0)[0]
    assert [int(x) for x in z] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = R.philox4(0xFFFFFFFFFFFFFFFF, np.array([0xFFFFFFFFFFFFFFFF], dtype=np.uint64), 0xFFFFFFFF, 0xFFFFFFFF)[0]
    assert [int(x) for x in f] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


def test_mask_rates_and_shapes():
    k = R.row_keep(7, R.RNG_IN_VID, 64, 2818, 0.5)
    assert k.shape == (64, 2818) and abs(float(k.mean()) - 0.5) < 0.01
    a = R.attn_keep(7, 1, 2, 4, 16, 0.1)
    assert a.shape == (2, 4, 16, 16) and abs(float(a.mean()) - 0.9) < 0.02
    s = R.droppath_scales(7, 4, 64, 0.1)
    assert s.shape == (4, 2, 64) and all(abs(v) < 1e-6 or abs(v - 1 / 0.9) < 1e-5 for v in np.unique(s).tolist())
