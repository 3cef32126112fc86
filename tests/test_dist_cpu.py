"""N>1 data-parallel path on CPU: world_size-2 gloo processes exercise the flat-buffer bucketed all-reduce /
broadcast used by univtg_amd.trainer, and check that averaged per-rank gradients equal the large-batch gradient
of the oracle for the rank-separable losses."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from univtg_amd.dist import allreduce_flat_, broadcast_flat_, shard_batch
    from oracle import univtg_oracle as O
    # 1. bucketed all-reduce on a flat buffer whose size is not a multiple of the bucket
    flat = torch.arange(1003, dtype=torch.float32) * (rank + 1)
    allreduce_flat_(flat, 256)
    ok1 = torch.equal(flat, torch.arange(1003, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    works = allreduce_flat_(flat.clone(), 100, async_op=True)
    for w in works:
        w.wait()
    # 2. parameter broadcast
    p = torch.full((77,), float(rank))
    broadcast_flat_(p, 0)
    ok2 = bool((p == 0).all())
    # 3. sharded gradient averaging == full-batch gradient (spans + labels losses are per-clip means over each rank's
    #    own shard with identical normalisers only when shards are balanced: use weighted recombination)
    cfg = O.make_cfg(hidden_dim=64, nheads=2, dim_feedforward=96, enc_layers=1, v_feat_dim=34, t_feat_dim=24, max_q_l=16,
                     input_dropout=0.0, dropout=0.0, droppath=0.0, losses=("labels",))
    params = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=1).items()}
    inputs, tg = O.make_batch(cfg, 4, 10, 6, seed=2, ragged=False)
    idx = shard_batch(4, rank, world)
    sub_in = {k: v[idx] for k, v in inputs.items()}
    sub_tg = {k: v[idx] for k, v in tg.items() if torch.is_tensor(v)}
    loss = O.criterion(O.forward(params, cfg, **sub_in), sub_tg, cfg)["loss_f"]
    loss.backward()
    g = torch.cat([p.grad.flatten() for p in params.values() if p.grad is not None])
    allreduce_flat_(g, 1 << 12)
    g /= world
    if rank == 0:
        params2 = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        full = O.criterion(O.forward(params2, cfg, **inputs), {k: v for k, v in tg.items() if torch.is_tensor(v)}, cfg)["loss_f"]
        full.backward()
        g2 = torch.cat([p.grad.flatten() for p in params2.values() if p.grad is not None])
        ok3 = bool(torch.allclose(g, g2, rtol=1e-4, atol=1e-6))      # unragged: every shard has the same number of valid clips
    else:
        ok3 = True
    q.put((rank, ok1, ok2, ok3))
    dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(all(r[1:]) for r in res), res


def test_gradient_bucket_ranges_partition_flat_buffer():
    """The overlap schedule of TrainStep (conv heads, then encoder layers last-to-first, then the rest) covers every element
    of the flat gradient buffer exactly once and matches the parameter table of include/uvtg.h."""
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace
    from univtg_amd.model import build_model
    from univtg_amd.trainer import TrainStep
    a = SimpleNamespace(device="cpu", hidden_dim=64, dropout=0.0, droppath=0.0, nheads=2, dim_feedforward=96, enc_layers=3,
                        dec_layers=2, pre_norm=False, position_embedding="sine", max_q_l=16, input_dropout=0.0, t_feat_dim=24,
                        v_feat_dim=34, span_loss_type="l1", use_txt_pos=False, n_input_proj=2, set_cost_span=10, set_cost_giou=1,
                        set_cost_class=4, max_v_l=75, b_loss_coef=10, g_loss_coef=1, f_loss_coef=10, s_loss_intra_coef=0.1,
                        s_loss_inter_coef=0.1, dset_type="vlp", train_path=["synthetic"], eos_coef=0.1, temperature=0.07,
                        saliency_margin=0.2)
    model, crit = build_model(a)
    step = TrainStep(model, crit)
    dims = model._dims(2, 5, 3, 34, 24, False)
    ranged, rest = step.bucket_ranges(dims)
    assert len(ranged) == model.enc_layers + 1
    cover = sorted(ranged + [r for r in rest if r[1] > r[0]])
    assert cover[0][0] == 0 and cover[-1][1] == step.flat.numel()
    assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
    # the first range is exactly span_embed + class_embed, the next ones whole encoder layers from the last to the first
    params = model._ordered_params()
    offs = model._offsets(dims)
    names = {id(p): k for k, p in model.named_parameters()}
    E = model.enc_layers
    head_ids = [i for i, p in enumerate(params) if names[id(p)].startswith(("span_embed", "class_embed"))]
    assert ranged[0] == (offs[min(head_ids)], offs[max(head_ids) + 1])
    for k, l in enumerate(range(E - 1, -1, -1)):
        ids = [i for i, p in enumerate(params) if names[id(p)].startswith(f"transformer.encoder.layers.{l}.")]
        assert ranged[1 + k] == (offs[min(ids)], offs[max(ids) + 1])
