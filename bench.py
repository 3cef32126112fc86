#!/usr/bin/env python
"""Benchmark of the UniVTG hot path on MI355X (BASELINE.json: clips/sec fwd+bwd, L=75, d=1024).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one full training step of BASELINE config 2 on synthetic features resident in HBM: input projections
-> 4-layer encoder -> conv heads + saliency -> dense criterion -> backward -> (RCCL gradient all-reduce) ->
global-norm clip + AdamW.  bf16 MFMA operands / fp32 accumulation, reference dropouts (0.5 / 0 / 0.1).
Per-GPU batch is fixed (weak scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOAD = dict(B=256, L_v=75, L_t=32, D_v=2818, D_t=512, d=1024, F=1024, H=8, E=4)


def model_args(**over):
    from types import SimpleNamespace
    a = dict(device="cuda", hidden_dim=WORKLOAD["d"], dropout=0.0, droppath=0.1, nheads=WORKLOAD["H"],
             dim_feedforward=WORKLOAD["F"], enc_layers=WORKLOAD["E"], dec_layers=2, pre_norm=False, position_embedding="sine",
             max_q_l=75, input_dropout=0.5, t_feat_dim=WORKLOAD["D_t"], v_feat_dim=WORKLOAD["D_v"], span_loss_type="l1",
             use_txt_pos=False, n_input_proj=2, set_cost_span=10, set_cost_giou=1, set_cost_class=4, max_v_l=75,
             b_loss_coef=10, g_loss_coef=1, f_loss_coef=10, s_loss_intra_coef=0.1, s_loss_inter_coef=0.1,
             dset_type="vlp", train_path=["synthetic"], eos_coef=0.1, temperature=0.07, saliency_margin=0.2)
    a.update(over)
    return SimpleNamespace(**a)


def synth_batch(B, Lv, Lt, Dv, Dt, seed, dev):
    """Synthetic config-2 batch generated ON DEVICE (shape/statistics of SURVEY 8d: L2-normalised feature blocks,
    TEF columns, ragged valid lengths, one GT window per sample with dense targets as main/dataset.py:173-230)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    lens_v = torch.randint(38, Lv + 1, (B,), generator=g, device=dev)
    lens_t = torch.randint(8, Lt + 1, (B,), generator=g, device=dev)
    lens_v[0], lens_t[0] = Lv, Lt
    tv = torch.arange(Lv, device=dev)[None]
    vm = (tv < lens_v[:, None]).float()
    tm = (torch.arange(Lt, device=dev)[None] < lens_t[:, None]).float()
    feat = torch.randn(B, Lv, Dv - 2, generator=g, device=dev)
    if Dv - 2 == 2816:
        feat = torch.cat([torch.nn.functional.normalize(feat[..., :2304], dim=-1), torch.nn.functional.normalize(feat[..., 2304:], dim=-1)], -1)
    else:
        feat = torch.nn.functional.normalize(feat, dim=-1)
    st = tv.float() / lens_v[:, None]
    vid = torch.cat([feat, st[..., None], (st + 1.0 / lens_v[:, None])[..., None]], -1) * vm[..., None]
    txt = torch.nn.functional.normalize(torch.randn(B, Lt, Dt, generator=g, device=dev), dim=-1) * tm[..., None]
    clip_len = 2.0
    ts = ((tv.float() + clip_len / 2) / lens_v[:, None]) * vm                    # dataset.py:173
    w0 = torch.rand(B, generator=g, device=dev) * 0.7
    ww = 0.05 + 0.25 * torch.rand(B, generator=g, device=dev)
    win = torch.stack([w0, torch.clamp(w0 + ww, max=1.0)], -1)                  # normalised GT window
    inside = ((ts >= win[:, None, 0]) & (ts <= win[:, None, 1]) & (vm > 0)).float()
    empty = inside.sum(1) == 0
    if bool(empty.any()):                                                       # dataset.py:202-205
        idx = torch.clamp((win[:, 0] * lens_v).long(), max=Lv - 1).clamp(min=0)
        inside[empty, idx[empty]] = 1.0
    span_nn = win[:, None, :] * inside[..., None]
    pos = torch.multinomial(inside + 1e-9, 1, generator=g)                      # dataset.py:230 (random fg clip)
    targets = dict(timestamp=torch.stack([ts, ts], -1).contiguous(), timestamp_mask=vm.contiguous(),
                   timestamp_window=inside.contiguous(), span_labels_nn=span_nn.contiguous(),
                   saliency_scores=inside.clone(), saliency_pos_labels=pos, _pos_idx=pos[:, 0].contiguous())
    inputs = dict(src_txt=txt.contiguous(), src_txt_mask=tm.contiguous(), src_vid=vid.contiguous(), src_vid_mask=vm.contiguous())
    # the valid lengths a collate knows on the host anyway (utils/tensor_utils.py:34-53 computes them to build the masks):
    # handing them over lets the engine run the packed (ragged) encoder stream without a device->host sync
    inputs["_lens_host"] = (lens_v.cpu().tolist(), lens_t.cpu().tolist())
    return inputs, targets


def cpu_baseline(sample_B=32, steps=3):
    """The oracle (CPU restatement of the reference, same torch CPU ops) timed on the host cores: fwd + criterion +
    bwd at the config-2 shape, bounded sample of `sample_B` samples per step."""
    from oracle import univtg_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))          # more threads than this only adds contention on the 2-socket host
    cfg = O.make_cfg(input_dropout=0.0, dropout=0.0, droppath=0.0)              # stochastic masks are not part of the oracle timing
    params = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=0).items()}
    inputs, tg = O.make_batch(cfg, sample_B, WORKLOAD["L_v"], WORKLOAD["L_t"], seed=0, ragged=True)
    times = []
    for i in range(steps + 1):
        for p in params.values():
            p.grad = None
        t0 = time.perf_counter()
        out = O.forward(params, cfg, **inputs)
        O.total_loss(O.criterion(out, tg, cfg), cfg).backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=sample_B * WORKLOAD["L_v"] / t, unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle fwd+criterion+bwd, fp32, B={sample_B} L_v=75 L_t=32 d=1024 E=4, median of {steps} steps after 1 warm-up "
                       f"({t:.2f} s/step)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=WORKLOAD["B"], help="per-GPU batch (config 2: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra instrumented steps for the roofline line")
    ap.add_argument("--no-padded-compare", action="store_true", help="skip the extra timing of the padded (non-packed) execution")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)", file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # plumbing test on a 1-GPU box: UVTG_BENCH_BACKEND=gloo UVTG_BENCH_SAME_DEVICE=1 runs the N>1 control flow with every rank on cuda:0
    backend = os.environ.get("UVTG_BENCH_BACKEND", "nccl")
    if os.environ.get("UVTG_BENCH_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=dev)     # "nccl" == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend=backend)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    from univtg_amd import _lib
    from univtg_amd.model import build_model
    from univtg_amd.trainer import TrainStep
    torch.manual_seed(2018)
    model, crit = build_model(model_args())
    model.to(dev).train()
    crit.to(dev).train()
    model.set_seed(2018 + rank)
    step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1)
    B, Lv, Lt = args.batch, WORKLOAD["L_v"], WORKLOAD["L_t"]
    batches = [synth_batch(B, Lv, Lt, WORKLOAD["D_v"], WORKLOAD["D_t"], 1000 * rank + i, dev) for i in range(2)]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step.step(*batches[i % 2])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step.step(*batches[i % 2])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    losses = step.losses[:5].tolist()

    # ---- the same batches through the padded execution (every padded position computed, as the reference does) ----
    padded_ms = None
    if rank == 0 and world == 1 and not args.no_padded_compare:
        step_p = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed=False)
        for i in range(3):
            step_p.step(*batches[i % 2])
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(10):
            step_p.step(*batches[i % 2])
        torch.cuda.synchronize()
        padded_ms = (time.perf_counter() - tp) / 10 * 1e3
        del step_p

    # ---- roofline of the dominant kernel: HIP events around every gemm_nt<bf16> launch, on the launch stream ----
    lib = _lib.load()
    roof = None
    if rank == 0:
        lib.uvtg_profile_start()
    for i in range(args.profile_steps):          # EVERY rank runs these steps (they contain the gradient all-reduce); only rank 0 instruments them
        step.step(*batches[i % 2])
    if rank == 0:
        ms, fl, n = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_longlong * 4)()
        _lib.check(lib.uvtg_profile_stop(ms, fl, n), "uvtg_profile_stop")
        fam = ["gemm_nt_kernel<bf16>", "gemm_nt_kernel<split-bf16>", "gemm_tn_kernel", "gemm_nt256_kernel"]
        dom = max(range(4), key=lambda i: ms[i])
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
        traffic, traffic_note = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_nt256.json")       # separate rocprofv3 --pmc passes of this same command
        if fam[dom] == "gemm_nt256_kernel" and os.path.exists(pmc):
            with open(pmc) as f:
                pj = json.load(f)
            traffic = pj["traffic_bytes_per_launch"]
            traffic_note = ("HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), committed in "
                            "profiles/r01_pmc_nt256.json; measured MFMA-busy fraction %.3f" % pj["mfma_busy_frac"])
        roof = dict(bound="mfma", kernel=fam[dom], achieved=round(ach, 2), peak=2500.0, unit="TFLOP/s", frac=round(ach / 2500.0, 4),
                    traffic=traffic, traffic_note=traffic_note, launches_per_step=int(n[dom] // max(1, args.profile_steps)),
                    avg_launch_us=round(ms[dom] * 1e3 / max(1, n[dom]), 2),
                    event_pair_floor_us=round(lib.uvtg_profile_event_floor_ms() * 1e3, 2),
                    algorithmic_gflop_per_launch=round(fl[dom] / max(1, n[dom]) / 1e9, 2),
                    all_gemm_kernels={fam[i]: dict(ms_per_step=round(ms[i] / max(1, args.profile_steps), 3),
                                                   tflops=round(fl[i] / (ms[i] * 1e-3) / 1e12, 1) if ms[i] > 0 else 0.0,
                                                   launches_per_step=int(n[i] // max(1, args.profile_steps))) for i in range(4)})
    if world > 1:
        torch.distributed.barrier()

    if rank == 0:
        clips = B * Lv * world * args.steps
        S, d, F_, E = Lv + Lt, WORKLOAD["d"], WORKLOAD["F"], WORKLOAD["E"]
        enc_flops = 3 * E * B * (8 * S * d * d + 4 * S * d * F_ + 4 * S * S * d)
        out = dict(metric="clips/sec (L=75,d=1024) fwd+bwd", value=round(clips / elapsed, 1), unit="clips/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                   config=dict(workload="QVHighlights training shape (BASELINE config 2): L_v=75 L_t=32 D_v=2818 D_t=512 d=1024 F=1024 "
                                        "H=8 E=4, full train step (fwd+criterion+bwd+clip+AdamW), dropout 0.5/0/0.1, ragged valid lengths (SURVEY 8d variant B), "
                                        "packed encoder stream (valid rows + one representative padded clip per sample)",
                               per_gpu_batch=B, global_batch=B * world, parallelism=f"dp{world}"),
                   samples_per_sec=round(B * world * args.steps / elapsed, 1),
                   encoder_mfma_frac_of_step=round(enc_flops / (elapsed / args.steps) / 2.5e15, 4),      # reference-algorithmic (padded) encoder FLOPs / step time / peak
                   packed_rows_fraction=round(sum(sum(a) + sum(b) + sum(1 for x in a if x < Lv) for a, b in (bt[0]["_lens_host"] for bt in batches)) / (len(batches) * B * (Lv + Lt)), 4),
                   padded_execution_ms_per_step=None if padded_ms is None else round(padded_ms, 3),
                   losses=[round(x, 5) for x in losses], roofline=roof, cpu_baseline=cpu)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
