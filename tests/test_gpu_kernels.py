"""Kernel-level parity on a real MI355X: every HIP kernel family against a plain torch fp32/fp64 statement
of the same op on the same seeded inputs (called through the C ABI via univtg_amd.ops)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16)


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("M,N,K,act", [(256, 256, 256, 0), (300, 200, 136, 1), (1024, 1024, 1024, 2), (77, 64, 96, 0),
                                       (27392, 1024, 1024, 0), (8292, 2056, 512, 2), (5000, 3080, 192, 1)])
def test_linear_bf16(dev, M, N, K, act):
    from univtg_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)     # asymmetric operands (transpose-detecting)
    b = torch.randn(N, generator=g).to(dev)
    ab, wb = bf(a), bf(w)
    got = ops.linear_bf16(ab, wb, b, act)
    ref = ab.double() @ wb.double().t() + b.double()
    ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref) if act == 2 else ref)
    assert relerr(got, ref) < 2e-5, relerr(got, ref)               # fp32 accumulation of exact bf16 products


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (130, 96, 2880), (515, 1024, 1024),
                                   (8192, 1024, 2818), (20158, 1024, 1024)])     # the last two: the persistent 256-wide kernel's split instantiations
def test_linear_f32x3(dev, M, N, K):
    from univtg_amd import ops
    g = torch.Generator().manual_seed(M * 7 + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    got = ops.linear_f32x3(a, w, b, 0)
    ref = a.double() @ w.double().t() + b.double()
    fp32 = a @ w.t() + b
    e3, e32 = relerr(got, ref), relerr(fp32, ref)
    print(f"\n[linear_f32x3 {M}x{N}x{K}] split-operand rel err {e3:.2e}, torch fp32 GEMM {e32:.2e}")
    assert e3 < 2e-6, (e3, e32)                                     # fp16 hi + lo images: ~2^-22 relative (the bf16 split of rounds 1-3: 2^-16)


@pytest.mark.parametrize("M,N,K,splits", [(256, 128, 128, 1), (1000, 136, 200, 3), (4096, 256, 2824, 4)])
def test_wgrad_tn(dev, M, N, K, splits):
    from univtg_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N)
    dy = bf(torch.randn(M, N, generator=g)).to(dev)
    x = bf(torch.randn(M, K, generator=g)).to(dev)
    dw, db = ops.wgrad_bf16(dy, x, splits)
    ref = dy.double().t() @ x.double()
    assert relerr(dw, ref) < 3e-5, relerr(dw, ref)
    assert relerr(db, dy.double().sum(0)) < 3e-5


@pytest.mark.parametrize("M,N,K", [(27392, 1024, 1024), (4100, 520, 2824), (19200, 1024, 2880), (2050, 256, 264)])
def test_wgrad_tn256(dev, M, N, K):
    """256-tile weight-gradient kernel (plain fp32 slabs + reduce): ragged M tail, ragged N / K tiles, bias gradient."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + 3 * N)
    dy = bf(torch.randn(M, N, generator=g)).to(dev)
    x = bf(torch.randn(M, K, generator=g)).to(dev)
    try:
        _lib.check(lib.uvtg_debug_force_nt_tile(256))
        dw, db = ops.wgrad_bf16_ws(dy, x)
    finally:
        lib.uvtg_debug_force_nt_tile(0)
    ref = dy.double().t() @ x.double()
    assert relerr(dw, ref) < 3e-5, relerr(dw, ref)
    assert relerr(db, dy.double().sum(0)) < 3e-5


@pytest.mark.parametrize("M,shapes", [
    (20158, [(1024, 1024)] * 4 + [(2048, 1024)] + [(1024, 1024)] * 3 + [(2048, 1024)] + [(1024, 1024)] * 7 + [(2048, 1024), (1024, 1024), (1024, 1024)] + [(2048, 1024)]),   # the encoder's 20 gradients: 384 tiles = 256 whole + 128 in halves
    (8200, [(1024, 1024)] * 5 + [(256, 1280)]),               # 85 tiles, fewer than CUs: every tile in 3 parts
    (27392, [(1024, 1024)] * 16),                             # 256 tiles: whole tiles only
    (4100, [(256, 256)] * 4 + [(3072, 1024)] * 2),            # 100 tiles in halves, ragged last 64-row step
    (27392, [(1024, 1024)] * 18)])                            # 288 tiles = 256 whole + 32 that would need 8 parts: declined
def test_wgrad_tn256_hybrid_multi(dev, M, shapes):
    """gemm_tn256h_kernel (several weight gradients over the same rows, whole tiles + split tiles folded by the last arriver, NO reduce
    pass) vs fp64.  Launched several times on CHANGING operands into NaN-filled outputs with fresh slabs: a stale slab line, a missed
    ticket or an unwritten output element cannot hide."""
    from univtg_amd import ops
    g = torch.Generator().manual_seed(M + len(shapes))
    for rep in range(3):
        dys = [bf(torch.randn(M, n, generator=g) * (1 + rep)).to(dev) for n, k in shapes]
        xs = [bf(torch.randn(M, k, generator=g)).to(dev) for n, k in shapes]
        try:
            dws, dbs = ops.wgrad_bf16_multi(dys, xs)
        except RuntimeError as e:
            assert "code -2" in str(e), e                       # declined: the launcher documents which splits it takes
            assert len(shapes) == 18                           # (the one parametrisation whose remainder tiles would need 8 parts)
            return
        for i, (n, k) in enumerate(shapes):
            ref = dys[i].double().t() @ xs[i].double()
            assert relerr(dws[i], ref) < 3e-5, (rep, i, relerr(dws[i], ref))
            assert relerr(dbs[i], dys[i].double().sum(0)) < 3e-5, (rep, i)


def test_wgrad_tn256_hybrid_under_uneven_load(dev):
    """The slab / ticket hand-off of the hybrid weight-gradient launch while another stream keeps part of the chip busy (uneven arrival of
    the parts of a tile; cdna_hip_programming.md Guideline 16: hand-offs must be tested under uneven load, not on an idle chip)."""
    from univtg_amd import ops
    M = 20158
    shapes = [(1024, 1024)] * 24
    g = torch.Generator().manual_seed(7)
    dys = [bf(torch.randn(M, n, generator=g)).to(dev) for n, k in shapes[:2]] * 12
    xs = [bf(torch.randn(M, k, generator=g)).to(dev) for n, k in shapes[:2]] * 12
    refs = [dys[i].double().t() @ xs[i].double() for i in range(2)]
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device=dev)
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(1 + rep):
                junk.mul_(1.0001)                                # HBM-bound traffic from another queue
        dws, dbs = ops.wgrad_bf16_multi(dys, xs)
        torch.cuda.synchronize()
        for i in range(24):
            assert relerr(dws[i], refs[i % 2]) < 3e-5, (rep, i)


@pytest.mark.parametrize("rows,D", [(64, 1024), (37, 2818), (50, 512), (9, 514), (33, 64), (5, 2817)])
def test_layernorm(dev, rows, D):
    from univtg_amd import ops
    g = torch.Generator().manual_seed(rows * D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).to(dev)
    gam = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    bet = (0.1 * torch.randn(D, generator=g)).to(dev)
    y, mean, rstd = ops.layernorm_fwd(x, gam, bet)
    xr = x.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    assert float((y - ref).abs().max()) < 2e-5
    go = torch.randn(rows, D, generator=g).to(dev)
    ref.backward(go.double())
    dx, dg, db = ops.layernorm_bwd(go, x, mean, rstd, gam)
    assert relerr(dx, xr.grad) < 1e-4
    assert relerr(dg, gr.grad) < 1e-4 and relerr(db, br.grad) < 1e-4


def _attn_ref(qkv, kvalid, B, S, H, hd):
    d = H * hd
    q, k, v = [t.view(B, S, H, hd).transpose(1, 2) for t in qkv.double().view(B, S, 3 * d).split(d, dim=-1)]
    sc = q @ k.transpose(-1, -2)
    sc = sc.masked_fill(~kvalid.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(sc, -1)
    return (p @ v).transpose(1, 2).reshape(B * S, d), torch.logsumexp(sc, -1)


def _kvalid(B, S, g, dev):
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    kv = (torch.arange(S)[None, :] < lens[:, None])
    kv[:, S - 3:] = True                                           # text-like tail of valid keys after the padded gap
    return kv.to(torch.uint8).to(dev)


@pytest.mark.parametrize("B,S,H,hd", [(3, 27, 2, 32), (2, 107, 4, 64), (2, 107, 8, 128), (1, 300, 2, 128)])
@pytest.mark.parametrize("precise", [False, True])
def test_attention_fwd(dev, B, S, H, hd, precise):
    from univtg_amd import ops
    g = torch.Generator().manual_seed(S + hd)
    qkv = torch.randn(B * S, 3 * H * hd, generator=g)
    qkv[:, : H * hd] *= hd ** -0.5
    kv = _kvalid(B, S, g, dev)
    x = qkv.to(dev) if precise else bf(qkv).to(dev)
    o, lse = ops.attention_fwd(x, kv, B, S, H, hd, precise)
    ref_o, ref_l = _attn_ref(x.float().cpu(), kv.cpu(), B, S, H, hd)
    tol = 3e-5 if precise else 1.5e-2                              # bf16 P and O rounding in the fast path
    assert float((o.float().cpu() - ref_o).abs().max()) < tol
    assert float((lse.cpu() - ref_l).abs().max()) < (1e-4 if precise else 2e-3)


def test_attention_softmax_rescale_branch(dev):
    """One key spikes far above the rest in the SECOND key tile: forces the online-softmax rescale."""
    from univtg_amd import ops
    B, S, H, hd = 1, 150, 1, 64
    g = torch.Generator().manual_seed(1)
    qkv = 0.1 * torch.randn(B * S, 3 * H * hd, generator=g)
    qkv[5, :hd] = 3.0
    qkv[100, hd:2 * hd] = 3.0                                       # q5 . k100 = 576
    kv = torch.ones(B, S, dtype=torch.uint8, device=dev)
    o, _ = ops.attention_fwd(qkv.to(dev), kv, B, S, H, hd, True)
    ref_o, _ = _attn_ref(qkv, kv.cpu(), B, S, H, hd)
    assert float((o.cpu() - ref_o).abs().max()) < 1e-4


@pytest.mark.parametrize("B,S,H,hd", [(2, 27, 2, 32), (2, 107, 2, 64), (2, 107, 4, 128), (1, 200, 2, 128), (2, 128, 2, 128), (3, 33, 1, 128),
                                       (2, 160, 8, 128), (1, 256, 2, 64), (2, 129, 2, 32), (1, 300, 2, 128)])   # 129..256: the 8-wave fused kernel; 300: split kernels
def test_attention_bwd(dev, B, S, H, hd):
    from univtg_amd import ops
    g = torch.Generator().manual_seed(S * hd)
    d = H * hd
    qkv = torch.randn(B * S, 3 * d, generator=g)
    qkv[:, :d] *= hd ** -0.5
    kv = _kvalid(B, S, g, dev)
    xb = bf(qkv).to(dev)
    o, lse = ops.attention_fwd(xb, kv, B, S, H, hd, False)
    do = bf(torch.randn(B * S, d, generator=g)).to(dev)
    dqkv = ops.attention_bwd(xb, kv, o, lse, do, 1.0, B, S, H, hd)
    xr = xb.double().cpu().requires_grad_(True)
    ref_o, _ = _attn_ref(xr, kv.cpu(), B, S, H, hd)
    ref_o.backward(do.double().cpu())
    ref = xr.grad
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        e = relerr(dqkv[:, sl].float().cpu(), ref[:, sl])
        assert e < 2.5e-2, (name, e)


@pytest.mark.parametrize("B,S,H,hd", [(2, 107, 4, 128), (1, 300, 2, 128), (2, 1232, 3, 128), (3, 517, 1, 128), (2, 64, 2, 128)])
def test_attention_fwd_dma_tile_loop_bit_identical(dev, B, S, H, hd):
    """head_dim-128 bf16 forward (round 5): K / V tiles by LDS-DMA into a double buffer, one barrier per tile (the dQ kernel's tile loop) against the
    register-staged single-buffer kernel -- same products, same online-softmax order: outputs and lse bit for bit; and against the fp64 reference."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(S + 3 * hd)
    d = H * hd
    qkv = torch.randn(B * S, 3 * d, generator=g)
    qkv[:, :d] *= hd ** -0.5
    kv = _kvalid(B, S, g, dev)
    xb = bf(qkv).to(dev)
    try:
        _lib.check(lib.uvtg_debug_attn_fwd_dma(0))
        o0, l0 = ops.attention_fwd(xb, kv, B, S, H, hd, False)
        _lib.check(lib.uvtg_debug_attn_fwd_dma(1))
        o1, l1 = ops.attention_fwd(xb, kv, B, S, H, hd, False)
    finally:
        lib.uvtg_debug_attn_fwd_dma(1)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    if S <= 600:
        ref_o, ref_l = _attn_ref(xb.float().cpu(), kv.cpu(), B, S, H, hd)
        assert float((o1.float().cpu() - ref_o).abs().max()) < 1.5e-2
        assert float((l1.cpu() - ref_l).abs().max()) < 2e-3


@pytest.mark.parametrize("B,S,H,hd", [(1, 300, 2, 128), (2, 1232, 3, 128), (3, 517, 1, 128)])
def test_attention_bwd_role_split_dkdv_bit_identical(dev, B, S, H, hd):
    """Long-sequence attention backward at head_dim 128 (round 5): dK / dV by the role-split kernel (score waves hand P / dS to product waves through
    LDS; two waves per SIMD) against the one-wave-per-SIMD kernel it replaces -- same products, same accumulation order over the query blocks:
    bit for bit; and both against fp64 autograd (S = 1232 is BASELINE config 4's sequence)."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(S + 7)
    d = H * hd
    qkv = torch.randn(B * S, 3 * d, generator=g)
    qkv[:, :d] *= hd ** -0.5
    kv = _kvalid(B, S, g, dev)
    xb = bf(qkv).to(dev)
    o, lse = ops.attention_fwd(xb, kv, B, S, H, hd, False)
    do = bf(torch.randn(B * S, d, generator=g)).to(dev)
    try:
        _lib.check(lib.uvtg_debug_attn_ws(0))
        old = ops.attention_bwd(xb, kv, o, lse, do, 1.0, B, S, H, hd)
        _lib.check(lib.uvtg_debug_attn_ws(1))
        new = ops.attention_bwd(xb, kv, o, lse, do, 1.0, B, S, H, hd)
    finally:
        lib.uvtg_debug_attn_ws(1)
    assert torch.equal(old, new)
    if S <= 600:
        xr = xb.double().cpu().requires_grad_(True)
        ref_o, _ = _attn_ref(xr, kv.cpu(), B, S, H, hd)
        ref_o.backward(do.double().cpu())
        for name, sl in (("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
            e = relerr(new[:, sl].float().cpu(), xr.grad[:, sl])
            assert e < 2.5e-2, (name, e)


def test_sine_position(dev):
    from oracle import univtg_oracle as O
    from univtg_amd import ops
    B, Lv, Lt, d = 4, 75, 9, 1024
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(5, Lv + 1, (B,), generator=g)
    vm = (torch.arange(Lv)[None] < lens[:, None]).float()
    tm = torch.ones(B, Lt)
    tm[0, 5:] = 0
    i = torch.arange(d, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(i, 2).int() / d)
    pos, kvalid = ops.sine_position(vm.to(dev), tm.to(dev), dim_t.to(dev))
    ref = O.sine_position(vm, d)
    assert float((pos.cpu() - ref).abs().max()) < 5e-6
    assert torch.equal(kvalid.cpu().bool(), torch.cat([vm, tm], 1).bool())


def test_nt_tile_sizes_agree(dev):
    """The persistent 256-tile GEMM and the 128-tile GEMM give the same results (same bf16 products, fp32 accumulation
    in the same K order) on ragged shapes, with every epilogue the kernel-level entry point exposes."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(7)
    try:
        for (M, N, K, act) in [(5000, 3080, 192, 1), (700, 520, 1024, 2), (27392, 1024, 1024, 0)]:
            a = bf(torch.randn(M, K, generator=g).to(dev))
            w = bf((torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev))
            b = torch.randn(N, generator=g).to(dev)
            _lib.check(lib.uvtg_debug_force_nt_tile(128))
            r128 = ops.linear_bf16(a, w, b, act)
            _lib.check(lib.uvtg_debug_force_nt_tile(256))
            r256 = ops.linear_bf16(a, w, b, act)
            assert float((r128 - r256).abs().max()) <= 1e-6 * float(r128.abs().max()), (M, N, K)
    finally:
        lib.uvtg_debug_force_nt_tile(0)


@pytest.mark.parametrize("precise", [False, True])
def test_nt_small_launches(dev, precise):
    """The single-tile variants of the persistent NT GEMM (launches of at most one 128 x 128 / 128 x 256 tile per CU: three-stage staging
    ring) give the persistent kernel's results bit for bit; with a workspace, launches of <= half as many tiles as CUs (inference batches)
    also split K: same products, the K range summed in <= 4 parts folded in part order -- within fp32 summation noise of the unsplit launch,
    bit-reproducible from call to call (the fold does not depend on which part arrives last), tickets left zero, and every launch the rule
    excludes unchanged."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    ws = ops.sk_workspace(dev)
    lin = ops.linear_f32x3 if precise else ops.linear_bf16
    g = torch.Generator().manual_seed(23)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    #       M     N     K   act  (batch 32 encoder GEMM; batch 1; the video projection's K; ragged M / N with a K tail part; text rows)
    shapes = [(3424, 1024, 1024, 0), (107, 1024, 1024, 2), (1200, 1024, 2880 if not precise else 2816, 1), (333, 520, 1344, 0), (1024, 1024, 512, 0)]
    try:
        for (M, N, K, act) in shapes:
            a = torch.randn(M, K, generator=g).to(dev)
            w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
            b = torch.randn(N, generator=g).to(dev)
            if not precise:
                a, w = bf(a), bf(w)
            ref = a.double() @ w.double().t() + b.double()
            ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref) if act == 2 else ref)
            tol = 2e-6 if precise else 2e-5
            _lib.check(lib.uvtg_debug_nt_small(0))
            persistent = lin(a, w, b, act)
            assert relerr(persistent, ref) < tol
            for mode in (1, 2):                                                            # 128 x 128 tiles where they fit / 128 x 256 tiles only
                _lib.check(lib.uvtg_debug_nt_small(mode))
                assert torch.equal(lin(a, w, b, act), persistent), (M, N, K, mode)          # the ring alone: bit-identical
                got = lin(a, w, b, act, sk_ws=ws)
                assert int(ws[:256].view(torch.int32).abs().sum()) == 0                   # tickets handed back zero
                assert relerr(got, ref) < tol, (M, N, K, mode, relerr(got, ref))
                if mode == 1:
                    parts = lib.uvtg_debug_nt_splitk_parts(M, N, (2 if precise else 1) * ((K + 63) // 64 * 64), 1, cus)
                    assert (parts >= 2) == (M != 3424), (M, N, K, parts)                    # 216 tiles of 128 x 128: no split
                    assert torch.equal(got, persistent) == (parts == 0), (M, N, K)          # (a split launch DID take another summation order)
                for _ in range(4):
                    assert torch.equal(lin(a, w, b, act, sk_ws=ws), got), (M, N, K, mode)   # arrival order does not matter
        # excluded launches: more tiles than CUs / the split switched off -> the unsplit results, bit for bit
        _lib.check(lib.uvtg_debug_nt_small(1))
        a = bf(torch.randn(20000, 1024, generator=g).to(dev)); w = bf((torch.randn(1024, 1024, generator=g) / 32).to(dev))
        assert lib.uvtg_debug_nt_splitk_parts(20000, 1024, 1024, 1, cus) == 0
        assert torch.equal(ops.linear_bf16(a, w), ops.linear_bf16(a, w, sk_ws=ws))
        _lib.check(lib.uvtg_debug_nt_splitk(0))
        assert torch.equal(ops.linear_bf16(a[:1024], w), ops.linear_bf16(a[:1024], w, sk_ws=ws))
    finally:
        _lib.check(lib.uvtg_debug_nt_splitk(4))
        _lib.check(lib.uvtg_debug_nt_small(1))


def test_nt_loader_waves_bit_identical(dev):
    """Staging by one wave per SIMD ("loader waves", the default at tile heights 128 - 256) and by every wave for itself give the same bits:
    same LDS image, same products, same K order (every specialised epilogue the kernel-level entry point reaches, every height)."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(41)
    try:
        _lib.check(lib.uvtg_debug_force_nt_tile(256))
        for (M, N, K, act) in [(19850, 1024, 1024, 0), (5000, 3080, 192, 1), (9000, 520, 1024, 2)]:
            a = bf(torch.randn(M, K, generator=g).to(dev))
            w = bf((torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev))
            b = torch.randn(N, generator=g).to(dev)
            for bm in (128, 192, 256):
                _lib.check(lib.uvtg_debug_force_nt_bm(bm))
                _lib.check(lib.uvtg_debug_nt_loader_waves(0))
                own = ops.linear_bf16(a, w, b, act)
                _lib.check(lib.uvtg_debug_nt_loader_waves(7))
                assert torch.equal(ops.linear_bf16(a, w, b, act), own), (M, N, K, bm)
    finally:
        lib.uvtg_debug_nt_loader_waves(7)
        lib.uvtg_debug_force_nt_bm(0)
        lib.uvtg_debug_force_nt_tile(0)


@pytest.mark.parametrize("D", [1024, 512])
def test_layernorm_fwd_bf16_lean_vs_generic_vs_fp32(dev, D):
    """The encoder's bf16 LayerNorm forward (round 5: ln_fwd_lean_kernel, next row in flight, gamma / beta in registers) against torch fp32 on the
    same bf16 rows and against the generic row kernel it replaces: y, y + pos (clip rows only), mean, rstd."""
    from univtg_amd import _lib
    from univtg_amd.ops import _ptr, _stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(7)
    B, S, Lv = 41, 107, 75
    rows = B * S
    x = bf((torch.randn(rows, D, generator=g) * 1.7 + 0.3).to(dev))
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    pos = torch.randn(B * Lv, D, generator=g).to(dev)
    res = {}
    try:
        for lean in (1, 0):
            _lib.check(lib.uvtg_debug_ln_fwd_lean(lean))
            y = torch.empty(rows, D, dtype=torch.bfloat16, device=dev); u = torch.empty_like(y)
            mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
            _lib.check(lib.uvtg_debug_layernorm_fwd_bf16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(u), _ptr(pos), S, Lv, _ptr(mean), _ptr(rstd), rows, D, _stream()))
            y1 = torch.empty_like(y)
            _lib.check(lib.uvtg_debug_layernorm_fwd_bf16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y1), None, None, 0, 0, None, None, rows, D, _stream()))
            torch.cuda.synchronize()
            assert torch.equal(y1, y)
            res[lean] = (y.float(), u.float(), mean, rstd)
    finally:
        lib.uvtg_debug_ln_fwd_lean(1)
    xf = x.float()
    ref = torch.nn.functional.layer_norm(xf, (D,), gamma, beta, 1e-5)
    refu = ref.clone().view(B, S, D)
    refu[:, :Lv] += pos.view(B, Lv, D)
    refu = refu.view(rows, D)
    for lean in (1, 0):
        y, u, mean, rstd = res[lean]
        assert float((y - ref).abs().max()) < 2.5e-2 and float((u - refu).abs().max()) < 4e-2          # bf16 rounding of |y| <= ~6
        assert float((y - ref).abs().mean()) < 2e-3
        assert float((mean - xf.mean(1)).abs().max()) < 1e-5
        assert float((rstd - 1 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)).abs().max()) < 1e-5
    # the two kernels round the same fp32 values: they may differ by one bf16 ulp where a value sits on a rounding boundary, not more
    d = (res[1][0] - res[0][0]).abs()
    assert float(d.max()) <= 2 ** -7 * 8 and float((d > 0).float().mean()) < 1e-2
    assert float((res[1][2] - res[0][2]).abs().max()) < 1e-6


def test_nt_column_group_order_and_forced_plans_bit_identical(dev):
    """Round-5 experiment knobs of the persistent NT GEMM change WHICH workgroup computes a tile and how a launch is cut into head + tail, never
    the products or the K order: column-group tile order (wide outputs: 12 column tiles walked in groups of 4 / 6 / 5), forced head / tail plans
    (tails of at most one tile per CU now take the single-tile variant) -- bit for bit against the default plan."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(43)
    try:
        for (M, N, K, act) in [(27392, 3072, 256, 0), (9000, 2100, 192, 2)]:
            a = bf(torch.randn(M, K, generator=g).to(dev))
            w = bf((torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev))
            b = torch.randn(N, generator=g).to(dev)
            _lib.check(lib.uvtg_debug_nt_cgw(0))
            base = ops.linear_bf16(a, w, b, act)
            for cgw in (4, 6, 5):
                _lib.check(lib.uvtg_debug_nt_cgw(cgw))
                assert torch.equal(ops.linear_bf16(a, w, b, act), base), (M, N, K, cgw)
            _lib.check(lib.uvtg_debug_nt_cgw(0))
            for plan in ((320, 320 * (M // 320 - 1), 192), (256, 256 * (M // 256 - 2), 128), (320, 0, 0), (192, 0, 0)):
                _lib.check(lib.uvtg_debug_nt_plan_override(M, N, *plan))
                assert torch.equal(ops.linear_bf16(a, w, b, act), base), (M, N, K, plan)
                _lib.check(lib.uvtg_debug_nt_cgw(6))
                assert torch.equal(ops.linear_bf16(a, w, b, act), base), (M, N, K, plan, "cgw 6")
                _lib.check(lib.uvtg_debug_nt_cgw(0))
            _lib.check(lib.uvtg_debug_nt_plan_override(0, 0, 0, 0, 0))
    finally:
        lib.uvtg_debug_nt_cgw(0)
        lib.uvtg_debug_nt_plan_override(0, 0, 0, 0, 0)


def test_nt256_tile_heights_agree(dev):
    """Every tile-height instantiation of the persistent GEMM gives identical results (same products, same K order)."""
    from univtg_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    try:
        _lib.check(lib.uvtg_debug_force_nt_tile(256))
        for (M, N, K, act) in [(19850, 1024, 1024, 0), (5000, 3080, 192, 1), (700, 520, 1024, 2)]:
            a = bf(torch.randn(M, K, generator=g).to(dev))
            w = bf((torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev))
            b = torch.randn(N, generator=g).to(dev)
            outs = []
            for bm in (256, 320, 192, 128):
                _lib.check(lib.uvtg_debug_force_nt_bm(bm))
                outs.append(ops.linear_bf16(a, w, b, act))
            ref = a.double() @ w.double().t() + b.double()
            ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref) if act == 2 else ref)
            assert relerr(outs[0], ref) < 2e-5
            for o in outs[1:]:
                assert float((o - outs[0]).abs().max()) <= 1e-6 * float(outs[0].abs().max()), (M, N, K)
    finally:
        lib.uvtg_debug_force_nt_bm(0)
        lib.uvtg_debug_force_nt_tile(0)
