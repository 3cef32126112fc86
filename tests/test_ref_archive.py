"""oracle/_ref (CPU-only): the build-time archive of the reference's model path that bench.py's cpu_baseline leg times
(`cpu_baseline.kind = "reference"`, north_star's "the reference's own PyTorch CPU forward ... on the same box").  Skipped where neither
/root/reference nor a shipped archive exists.  Runs in a subprocess: the archive provides top-level packages `model` / `utils`, which must
not leak into (or collide inside) the test process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import json, sys, torch
sys.path.insert(0, %r)
from oracle.build_ref import build_ref, import_ref_model, MEMBERS
assert build_ref(verbose=False), "no archive"
ref, manifest = import_ref_model()
assert sorted(manifest["members"]) == sorted(MEMBERS)
assert ".zip" in ref.__file__, ref.__file__
from oracle import univtg_oracle as O
from oracle.make_golden import ref_args
cfg = O.make_cfg(hidden_dim=64, nheads=4, dim_feedforward=64, enc_layers=2, v_feat_dim=34, t_feat_dim=32,
                 input_dropout=0.0, dropout=0.0, droppath=0.0)
params = O.init_params(cfg, seed=5)
inputs, tg = O.make_batch(cfg, 3, 12, 6, seed=6, ragged=True)
model, crit = ref.build_model(ref_args(cfg))
model.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
model.eval(); crit.eval()
out = model(**inputs)
want = O.forward(params, cfg, **inputs)
ld = crit(out, tg)
wl = O.criterion(want, tg, cfg)
err = {k: float((out[k] - want[k]).abs().max()) for k in ("pred_logits", "pred_spans", "saliency_scores")}
lerr = {k: abs(float(ld[k]) - float(wl[k])) for k in wl}
print(json.dumps(dict(err=err, lerr=lerr)))
"""


def test_reference_archive_is_the_reference_and_agrees_with_the_oracle():
    from oracle.build_ref import ARCHIVE, MEMBERS, REF
    if not (os.path.exists(ARCHIVE) or all(os.path.exists(os.path.join(REF, m)) for m in MEMBERS)):
        pytest.skip("no /root/reference and no shipped oracle/_ref archive")
    r = subprocess.run([sys.executable, "-c", _PROBE % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert max(rep["err"].values()) < 1e-5, rep
    assert max(rep["lerr"].values()) < 1e-5, rep


def test_archive_is_not_tracked_and_not_visible_to_the_product():
    """`oracle/_ref/` is a build artefact (git-ignored) and nothing under univtg_amd/ may reference it."""
    with open(os.path.join(ROOT, ".gitignore")) as f:
        assert "oracle/_ref/" in f.read().split()
    for dp, _, fs in os.walk(os.path.join(ROOT, "univtg_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dp, f), errors="replace") as fh:
                    assert "_ref" not in fh.read().replace("x_ref", "").replace("_reference", ""), f
