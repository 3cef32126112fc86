"""Thin tensor-level wrappers over the kernel-level entry points of include/uvtg.h (the same kernels the
engine launches).  Used by the parity tests and handy for experiments; no computation happens in Python."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .model import _f32c, _ptr, _stream


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("univtg_amd.ops: tensors must be on a ROCm device (no CPU fallback)")


def to_bf16_bits(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round-to-nearest-even) on device, returned as torch.bfloat16."""
    _need_cuda(x)
    x = _f32c(x)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().uvtg_cast_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "uvtg_cast_bf16")
    return out


def sk_workspace(device) -> torch.Tensor:
    """Split-K workspace of the small-launch GEMM entry points (tickets zeroed; include/uvtg.h: uvtg_linear_*_sk)."""
    return torch.zeros(int(_lib.load().uvtg_linear_sk_ws_floats()), device=device)


def linear_bf16(a_bf16, w_bf16, bias=None, act=0, sk_ws=None):
    """C = act(A W^T + b): A [M,K] bf16, W [N,K] bf16 -> fp32 [M,N].  sk_ws (sk_workspace()): small launches may split K."""
    _need_cuda(a_bf16)
    M, K = a_bf16.shape
    N = w_bf16.shape[0]
    out = torch.empty(M, N, device=a_bf16.device)
    if sk_ws is not None:
        _lib.check(_lib.load().uvtg_linear_bf16_sk(_ptr(a_bf16.contiguous()), _ptr(w_bf16.contiguous()), _ptr(bias), _ptr(out),
                                                   M, N, K, act, _ptr(sk_ws), _stream()), "uvtg_linear_bf16_sk")
        return out
    _lib.check(_lib.load().uvtg_linear_bf16(_ptr(a_bf16.contiguous()), _ptr(w_bf16.contiguous()), _ptr(bias), _ptr(out),
                                            M, N, K, act, _stream()), "uvtg_linear_bf16")
    return out


def split_f16(x, is_weight=False):
    """fp16 hi | lo operand images [rows, 2 * kp] of an fp32 matrix (kp = columns rounded up to 64); see include/uvtg.h."""
    _need_cuda(x)
    x = _f32c(x)
    rows, cols = x.shape
    kp = (cols + 63) // 64 * 64
    out = torch.empty(rows, 2 * kp, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().uvtg_split_f16(_ptr(x), _ptr(out), rows, cols, kp, int(is_weight), _stream()), "uvtg_split_f16")
    return out


def linear_f32x3(a, w, bias=None, act=0, sk_ws=None):
    """nn.Linear in the precise arithmetic: fp32 operands -> fp16 hi | lo images -> three-product split GEMM (uvtg_linear_split)."""
    _need_cuda(a)
    a, w = _f32c(a), _f32c(w)
    M, K = a.shape
    N = w.shape[0]
    sa, sw = split_f16(a, False), split_f16(w, True)
    out = torch.empty(M, N, device=a.device)
    if sk_ws is not None:
        _lib.check(_lib.load().uvtg_linear_split_sk(_ptr(sa), _ptr(sw), _ptr(bias), _ptr(out), M, N, sa.shape[1] // 2, act, _ptr(sk_ws), _stream()),
                   "uvtg_linear_split_sk")
        return out
    _lib.check(_lib.load().uvtg_linear_split(_ptr(sa), _ptr(sw), _ptr(bias), _ptr(out), M, N, sa.shape[1] // 2, act, _stream()), "uvtg_linear_split")
    return out


def wgrad_bf16(dy_bf16, x_bf16, splits=4, with_bias=True):
    """dW [N,K] = dY^T X, db [N] = colsum(dY)."""
    _need_cuda(dy_bf16)
    M, N = dy_bf16.shape
    K = x_bf16.shape[1]
    dw = torch.zeros(N, K, device=dy_bf16.device)
    db = torch.zeros(N, device=dy_bf16.device) if with_bias else None
    _lib.check(_lib.load().uvtg_wgrad_bf16(_ptr(dy_bf16.contiguous()), _ptr(x_bf16.contiguous()), _ptr(dw), _ptr(db),
                                           M, N, K, splits, _stream()), "uvtg_wgrad_bf16")
    return dw, db


def wgrad_bf16_ws(dy_bf16, x_bf16, with_bias=True):
    """Same as wgrad_bf16 with the scratch that enables the 256-tile slab + reduce kernel."""
    _need_cuda(dy_bf16)
    lib = _lib.load()
    M, N = dy_bf16.shape
    K = x_bf16.shape[1]
    dw = torch.zeros(N, K, device=dy_bf16.device)
    db = torch.zeros(N, device=dy_bf16.device) if with_bias else None
    nf = lib.uvtg_wgrad_scratch_floats(M, N, K)
    scratch = torch.empty(nf, device=dy_bf16.device)
    _lib.check(lib.uvtg_wgrad_bf16_ws(_ptr(dy_bf16.contiguous()), _ptr(x_bf16.contiguous()), _ptr(dw), _ptr(db), M, N, K,
                                      _ptr(scratch), nf, _stream()), "uvtg_wgrad_bf16_ws")
    return dw, db


def wgrad_bf16_multi(dys, xs, with_bias=True):
    """Several weight gradients over the same rows in one launch, no reduce pass (uvtg_wgrad_bf16_multi).  dys[i] [M, N_i], xs[i] [M, K_i]
    bf16-bit tensors; returns ([dW_i], [dbias_i])."""
    lib = _lib.load()
    _need_cuda(dys[0])
    n = len(dys)
    M = dys[0].shape[0]
    dev = dys[0].device
    Ns, Ks = [int(t.shape[1]) for t in dys], [int(t.shape[1]) for t in xs]
    tiles = sum((a // 256) * (b // 256) for a, b in zip(Ns, Ks))
    nf = lib.uvtg_wgrad_multi_slab_floats(tiles)
    slabs = torch.empty(nf, device=dev)
    tickets = torch.zeros(tiles, dtype=torch.int32, device=dev)
    dws = [torch.full((a, b), float("nan"), device=dev) for a, b in zip(Ns, Ks)]       # assigned: every element must be written
    dbs = [torch.zeros(a, device=dev) for a in Ns] if with_bias else None
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    iarr = lambda vs: (C.c_int * n)(*vs)
    _lib.check(lib.uvtg_wgrad_bf16_multi(n, arr(dys), iarr(Ns), arr(xs), iarr(Ks), arr(dws), arr(dbs) if with_bias else None, M,
                                         _ptr(slabs), nf, _ptr(tickets), tiles, _stream()), "uvtg_wgrad_bf16_multi")
    return dws, dbs


def layernorm_fwd(x, gamma, beta):
    _need_cuda(x)
    x = _f32c(x)
    rows, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device)
    rstd = torch.empty(rows, device=x.device)
    _lib.check(_lib.load().uvtg_layernorm_fwd(_ptr(x), _ptr(_f32c(gamma)), _ptr(_f32c(beta)), _ptr(y), _ptr(mean), _ptr(rstd),
                                              rows, D, _stream()), "uvtg_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(g, x, mean, rstd, gamma):
    _need_cuda(x)
    rows, D = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.zeros(D, device=x.device)
    dbeta = torch.zeros(D, device=x.device)
    _lib.check(_lib.load().uvtg_layernorm_bwd(_ptr(_f32c(g)), _ptr(_f32c(x)), _ptr(mean), _ptr(rstd), _ptr(_f32c(gamma)),
                                              _ptr(dx), _ptr(dgamma), _ptr(dbeta), rows, D, _stream()), "uvtg_layernorm_bwd")
    return dx, dgamma, dbeta


def attention_fwd(qkv, kvalid, B, S, H, hd, precise=False):
    """qkv [B*S, 3*H*hd] (bf16, or fp32 when precise; q already scaled), kvalid [B,S] uint8 -> o [B*S, H*hd], lse [B,H,S]."""
    _need_cuda(qkv)
    o = torch.empty(B * S, H * hd, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, S, device=qkv.device)
    _lib.check(_lib.load().uvtg_attention_fwd(_ptr(qkv.contiguous()), _ptr(kvalid.contiguous()), _ptr(o), _ptr(lse),
                                              B, S, H, hd, int(precise), _stream()), "uvtg_attention_fwd")
    return o, lse


def attention_bwd(qkv, kvalid, o, lse, do, qscale, B, S, H, hd):
    _need_cuda(qkv)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B, H, S, device=qkv.device)
    _lib.check(_lib.load().uvtg_attention_bwd(_ptr(qkv.contiguous()), _ptr(kvalid.contiguous()), _ptr(o.contiguous()), _ptr(lse),
                                              _ptr(do.contiguous()), _ptr(delta), _ptr(dqkv), float(qscale), B, S, H, hd, _stream()),
               "uvtg_attention_bwd")
    return dqkv


def sine_position(vid_mask, txt_mask, dim_t):
    _need_cuda(vid_mask)
    B, Lv = vid_mask.shape
    Lt = txt_mask.shape[1]
    d = dim_t.numel()
    pos = torch.empty(B, Lv, d, device=vid_mask.device)
    kvalid = torch.empty(B, Lv + Lt, dtype=torch.uint8, device=vid_mask.device)
    _lib.check(_lib.load().uvtg_sine_position(_ptr(_f32c(vid_mask)), _ptr(_f32c(txt_mask)), _ptr(_f32c(dim_t)), _ptr(pos), _ptr(kvalid),
                                              B, Lv, Lt, d, _stream()), "uvtg_sine_position")
    return pos, kvalid


def decode_rank_nms(pred_logits, pred_spans, timestamp, timestamp_mask, durations, nms_thd=0.7, max_before=1000, max_after=10):
    """Device version of main/inference_mr.py:109-160 + utils/temporal_nms.py.  Returns (windows [B,Lv,3] float64 ranked
    rows, order [B,Lv] int32 clip index per rank, keep [B,max_after] int32 ranked positions kept (-1 pad), n_keep [B])."""
    _need_cuda(pred_logits)
    B, Lv = pred_logits.shape[:2]
    dev = pred_logits.device
    win = torch.empty(B, Lv, 3, dtype=torch.float64, device=dev)
    order = torch.empty(B, Lv, dtype=torch.int32, device=dev)
    keep = torch.empty(B, max_after, dtype=torch.int32, device=dev)
    nk = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().uvtg_decode_rank_nms(_ptr(_f32c(pred_logits)), _ptr(_f32c(pred_spans)), _ptr(_f32c(timestamp)),
                                                _ptr(_f32c(timestamp_mask)), _ptr(_f32c(durations)), B, Lv, float(nms_thd),
                                                int(max_before), int(max_after), _ptr(win), _ptr(order), _ptr(keep), _ptr(nk),
                                                _stream()), "uvtg_decode_rank_nms")
    return win, order, keep, nk


def postprocess_mr(pred_logits, pred_spans, saliency, timestamp, timestamp_mask, durations, clip_length=0.0, eval_mode="add",
                   nms_thd=0.7, max_before=1000, max_after=10):
    """decode_rank_nms + PostProcessorDETR round_multiple (clip_length > 0) before the NMS + pred_saliency_scores
    (fp16(saliency) [+ prob]); main/inference_mr.py:109-192, eval/postprocessing.py:46-51.  Returns
    (windows, order, keep, n_keep, saliency_out[B,Lv] fp32 or None)."""
    _need_cuda(pred_logits)
    B, Lv = pred_logits.shape[:2]
    dev = pred_logits.device
    win = torch.empty(B, Lv, 3, dtype=torch.float64, device=dev)
    order = torch.empty(B, Lv, dtype=torch.int32, device=dev)
    keep = torch.empty(B, max_after, dtype=torch.int32, device=dev)
    nk = torch.empty(B, dtype=torch.int32, device=dev)
    sal_out = torch.empty(B, Lv, device=dev) if saliency is not None else None
    _lib.check(_lib.load().uvtg_postprocess_mr(_ptr(_f32c(pred_logits)), _ptr(_f32c(pred_spans)), _ptr(None if saliency is None else _f32c(saliency)),
                                               _ptr(_f32c(timestamp)), _ptr(_f32c(timestamp_mask)), _ptr(_f32c(durations)), B, Lv,
                                               float(clip_length), int(eval_mode == "add"), float(nms_thd), int(max_before), int(max_after),
                                               _ptr(win), _ptr(order), _ptr(keep), _ptr(nk), _ptr(sal_out), _stream()), "uvtg_postprocess_mr")
    return win, order, keep, nk, sal_out
