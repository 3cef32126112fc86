"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- times the REAL reference (showlab/UniVTG, /root/reference) on the CPU cores of the BUILD
container: model/univtg.py build_model() -> Model.forward + SetCriterion + backward, train mode, BASELINE config 2 (B=256, L_v=75, L_t=32,
d=1024, E=4), on the same synthetic batch generator bench.py's cpu_baseline leg uses.

    python oracle/time_reference_cpu.py [steps] > profiles/r04_reference_cpu_container.json

north_star asks for "the reference's own PyTorch CPU forward timed on the host cores of the same box"; /root/reference does not exist on the GPU
box (it cannot be vendored), so bench.py times a port built from the same torch.nn modules there (cpu_baseline.kind = "port",
oracle/nn_baseline.py, pinned to the oracle).  This script is the other half of the evidence: the reference itself and that port, timed
side by side HERE, on one machine -- the port is an honest stand-in if the two agree."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import univtg_oracle as O            # noqa: E402
from oracle.make_golden import import_reference, ref_args   # noqa: E402
from oracle.nn_baseline import NNBaseline        # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    ref = import_reference()
    cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=0)
    inputs, tg = O.make_batch(cfg, 256, 75, 32, seed=1, ragged=False)
    B, Lv = inputs["src_vid"].shape[:2]
    out = {}
    # ---- the reference itself ----
    model, crit = ref.univtg.build_model(ref_args(cfg))
    model.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    model.train(); crit.train()

    def ref_step():
        model.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        o = model(**inputs)
        t1 = time.perf_counter()
        ld = crit(o, tg)
        sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict).backward()
        return time.perf_counter() - t0, t1 - t0
    ref_step()
    ts = [ref_step() for _ in range(steps)]
    out["reference"] = dict(step_s=[round(a, 3) for a, _ in ts], forward_s=[round(b, 3) for _, b in ts],
                            clips_per_sec_fwd_bwd=round(steps * B * Lv / sum(a for a, _ in ts), 1),
                            clips_per_sec_forward_only=round(steps * B * Lv / sum(b for _, b in ts), 1))
    # ---- the port bench.py times on the GPU box ----
    port = NNBaseline(cfg)
    port.load_state_dict(params, strict=True)
    port.train()

    def port_step():
        port.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        o = port(**inputs)
        t1 = time.perf_counter()
        O.total_loss(O.criterion(o, tg, cfg), cfg).backward()
        return time.perf_counter() - t0, t1 - t0
    port_step()
    tp = [port_step() for _ in range(steps)]
    out["port_nn_modules"] = dict(step_s=[round(a, 3) for a, _ in tp], forward_s=[round(b, 3) for _, b in tp],
                                  clips_per_sec_fwd_bwd=round(steps * B * Lv / sum(a for a, _ in tp), 1),
                                  clips_per_sec_forward_only=round(steps * B * Lv / sum(b for _, b in tp), 1))
    out["meta"] = dict(what="showlab/UniVTG Model.forward + SetCriterion + backward (train mode, input dropout 0.5, DropPath 0.1) vs oracle/nn_baseline.py, "
                            "BASELINE config 2 at B=256, all-ones masks, fp32", threads=threads, host_cpus=os.cpu_count(), torch=torch.__version__,
                       steps=steps, where="build container (no GPU); /root/reference is not shipped to the GPU box",
                       ratio_port_over_reference=round(out["port_nn_modules"]["clips_per_sec_fwd_bwd"] / out["reference"]["clips_per_sec_fwd_bwd"], 3))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
