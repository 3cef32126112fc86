"""Test-side restatement (numpy) of the counter-based dropout stream the HIP kernels draw from
(univtg_amd/csrc/uvtg_common.h: philox4 / u01 / UVTG_RNG_*), so that train-mode runs on the GPU can be replayed
through the CPU oracle with the SAME Bernoulli masks (oracle/univtg_oracle.py: forward(rng=...)).

The reference draws its masks from torch's generator (nn.Dropout, drop_path: model/transformer_encoder_droppath.py:154-183);
mask VALUES are therefore not comparable, only the semantics given a mask -- which is what these helpers pin."""
import numpy as np

RNG_IN_VID, RNG_IN_TXT, RNG_ATTN, RNG_PATH, RNG_TXT_POS = 0x100, 0x200, 0x300, 0x400, 0x500
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)


def philox4(seed: int, ctr_lo: np.ndarray, stream: int, c3: int = 0x5EED5EED) -> np.ndarray:
    """Philox-4x32-10 keyed by the 64-bit seed; counter = (ctr_lo[31:0], ctr_lo[63:32], stream, 0x5eed5eed).
    Returns uint32 array of shape ctr_lo.shape + (4,)."""
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c0, c1 = ctr_lo & _LO, ctr_lo >> np.uint64(32)
    c2 = np.full_like(c0, np.uint64(stream))
    c3 = np.full_like(c0, np.uint64(c3))
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & _LO
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & _LO
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def u01(x: np.ndarray) -> np.ndarray:
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def row_keep(seed: int, stream: int, rows: int, D: int, p: float) -> np.ndarray:
    """keep mask (rows, D) of the dropout fused into the LayerNorm kernels (norm.hip: drop_mask4):
    one Philox call per 4 consecutive columns, counter = row * ceil(D/4) + col/4."""
    D4 = (D + 3) // 4
    ctr = np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(D4) + np.arange(D4, dtype=np.uint64)[None, :]
    u = u01(philox4(seed, ctr, stream)).reshape(rows, D4 * 4)[:, :D]
    return (u >= np.float32(p)).astype(np.float32)


def attn_keep(seed: int, layer: int, B: int, H: int, S: int, p: float) -> np.ndarray:
    """keep mask (B, H, S, S) of the attention-probability dropout (attn.hip: keep_scale)."""
    ctr = np.arange(B * H * S * S, dtype=np.uint64)
    u = u01(philox4(seed, ctr, RNG_ATTN + layer)[..., 0])
    return (u >= np.float32(p)).astype(np.float32).reshape(B, H, S, S)


def droppath_scales(seed: int, E: int, B: int, p: float) -> np.ndarray:
    """(E, 2, B) per-sample residual-branch factors in {0, 1/keep} (misc.hip: droppath_kernel)."""
    keep = np.float32(1.0 - p)
    u = u01(philox4(seed, np.arange(2 * E * B, dtype=np.uint64), RNG_PATH)[..., 0])
    return (np.floor(keep + u) / keep).astype(np.float32).reshape(E, 2, B)
