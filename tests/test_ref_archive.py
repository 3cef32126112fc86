"""oracle/_ref (CPU-only): the build-time archive of the reference's model + post-processing path that bench.py's cpu_baseline leg times
(`cpu_baseline.kind = "reference"`, north_star's "the reference's own PyTorch CPU forward ... on the same box") and that the GPU suite puts
beside the HIP path at production width.  Skipped where neither /root/reference nor a shipped archive exists.  The archive is only ever
imported in a child process (oracle/ref_runner.py): it provides top-level packages `model` / `utils` / `eval`, which must not leak into (or
collide inside) the test process."""
import json
import os
import sys
import zipfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_archive():
    from oracle.build_ref import ARCHIVE, MEMBERS, REF, build_ref
    if not (os.path.exists(ARCHIVE) or all(os.path.exists(os.path.join(REF, m)) for m in MEMBERS)):
        pytest.skip("no /root/reference and no shipped oracle/_ref archive")
    assert build_ref(verbose=False)
    return ARCHIVE


def test_reference_archive_is_the_reference_and_agrees_with_the_oracle(tmp_path):
    from oracle import univtg_oracle as O
    from oracle.build_ref import MEMBERS
    from oracle.ref_runner import run_job
    archive = _have_archive()
    with zipfile.ZipFile(archive) as z:
        names = set(z.namelist())
        manifest = json.loads(z.read("MANIFEST.json"))
        assert b"MIT License" in z.read("LICENSE")                     # the reference's licence travels with its files
    assert sorted(manifest["members"]) == sorted(MEMBERS) and set(MEMBERS) <= names
    cfgd = dict(hidden_dim=64, nheads=4, dim_feedforward=64, enc_layers=2, v_feat_dim=34, t_feat_dim=32, input_dropout=0.0, dropout=0.0, droppath=0.0)
    ev = str(tmp_path / "ev.npz")
    r = run_job(dict(task="model", threads=4, cfg=cfgd, param_seed=5, batch=dict(B=3, Lv=12, Lt=6, seed=6, ragged=True), eval_out=ev, train_steps=1))
    assert ".zip" in r["module_file"] and len(r["train_step_s"]) == 1
    cfg = O.make_cfg(**cfgd)
    params = O.init_params(cfg, seed=5)
    inputs, tg = O.make_batch(cfg, 3, 12, 6, seed=6, ragged=True)
    with torch.no_grad():
        want = O.forward(params, cfg, **inputs)
    got = np.load(ev)
    for k in ("pred_logits", "pred_spans", "saliency_scores"):
        assert float(np.abs(got[k] - want[k].numpy()).max()) < 1e-5, k
    # the reference's own inference tail (round_multiple + temporal_nms behind the compose glue) vs the restated post-processing oracle
    from oracle import postproc_oracle as P
    durations = np.array([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(3)])
    pin, pj = str(tmp_path / "pp.npz"), str(tmp_path / "pp.json")
    np.savez(pin, pred_logits=got["pred_logits"], pred_spans=got["pred_spans"], timestamp=tg["timestamp"].numpy(),
             timestamp_mask=tg["timestamp_mask"].numpy(), durations=durations)
    r2 = run_job(dict(task="postproc", outputs_npz=pin, result_json=pj, nms_thd=0.5, max_after_nms=4))
    assert all(".zip" in f for f in r2["module_files"])
    with open(pj) as f:
        tail = json.load(f)
    pre = P.decode_windows(got["pred_logits"], got["pred_spans"], tg["timestamp"].numpy(), tg["timestamp_mask"].numpy(), durations.tolist())
    for cl in (0.0, 2.0):
        p = pre if cl == 0 else [P.round_multiple(q, cl) for q in pre]
        assert p == tail[str(cl)]["pre"]
        assert [P.temporal_nms(q[:1000], 0.5, 4) for q in p] == tail[str(cl)]["nms"]


def test_a_shadowed_reference_import_is_refused(tmp_path):
    """A foreign top-level `model` package ahead of the archive must make the child fail loudly, not time something else."""
    import subprocess
    _have_archive()
    (tmp_path / "model").mkdir()
    (tmp_path / "model" / "__init__.py").write_text("")
    (tmp_path / "model" / "univtg.py").write_text("def build_model(a):\n    raise SystemExit('shadow')\n")
    code = (f"import sys; sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(0, {ROOT!r}); import model\n"
            "from oracle.build_ref import import_ref\n"
            "try:\n    import_ref(('model.univtg',))\nexcept ImportError as e:\n    print('REFUSED', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_archive_is_not_tracked_and_not_visible_to_the_product():
    """`oracle/_ref/` is a build artefact (git-ignored) and nothing under univtg_amd/ may reference it."""
    with open(os.path.join(ROOT, ".gitignore")) as f:
        assert "oracle/_ref/" in f.read().split()
    for dp, _, fs in os.walk(os.path.join(ROOT, "univtg_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dp, f), errors="replace") as fh:
                    assert "_ref" not in fh.read().replace("x_ref", "").replace("_reference", ""), f
