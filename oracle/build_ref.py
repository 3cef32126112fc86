"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- build-time recipe for `oracle/_ref/` (git-ignored; it travels to the GPU box with the
built tree exactly like `univtg_amd/libuvtg.so` does).

north_star: "the reference's own PyTorch CPU forward is timed on the host cores of the same box in the same run as the reported (not
optimised-against) baseline".  `/root/reference` exists only in the build container, so `__graft_entry__.build()` -- which runs there --
packs the files of the reference's model path AND of its post-processing path, unmodified and straight from where they lie, into ONE archive

    oracle/_ref/uvtg_reference_model.zip        model/univtg.py  (build_model, Model.forward, SetCriterion: model/univtg.py:105-155,195-351,409-450)
                                                model/transformer_encoder_droppath.py  model/position_encoding.py  model/matcher.py
                                                utils/span_utils.py
                                                utils/temporal_nms.py  (temporal_nms: utils/temporal_nms.py:25-74)
                                                eval/postprocessing.py (PostProcessorDETR.round_to_multiple_clip_lengths: eval/postprocessing.py:46-51)
                                                eval/eval.py  eval/utils.py  utils/basic_utils.py   (what eval/postprocessing.py imports at module level)
                                                LICENSE (the reference's MIT licence and copyright notice travel with its files)
                                                (+ empty package markers, + MANIFEST.json with the sha256 of every member)

which is imported ONLY inside a child process (`oracle/ref_runner.py`, started by `bench.py`'s `cpu_baseline` leg and by the tests): the archive
provides the generic top-level packages `model` / `utils` / `eval`, and a child process that has nothing else on its import path cannot
shadow them or be shadowed (`import_ref_model` verifies that every module it returns was loaded FROM the archive).  `cpu_baseline.kind =
"reference"`; only when the archive is absent does the bench fall back to the port (`oracle/nn_baseline.py`, `kind = "port"`) and say so.
No reference source enters the repository: the archive is a build artefact like the .so, `oracle/_ref/` is listed in .gitignore, and nothing
in the product path (`univtg_amd/`) can see it.

    python oracle/build_ref.py            # (re)build when /root/reference exists; no-op (keeps a shipped archive) otherwise
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("UVTG_REFERENCE", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "uvtg_reference_model.zip")
MEMBERS = ["model/univtg.py", "model/transformer_encoder_droppath.py", "model/position_encoding.py", "model/matcher.py",
           "utils/span_utils.py", "utils/temporal_nms.py", "eval/postprocessing.py", "eval/eval.py", "eval/utils.py", "utils/basic_utils.py",
           "LICENSE"]
PACKAGES = ("model", "utils", "eval")
_EPOCH = (1980, 1, 1, 0, 0, 0)          # fixed member timestamps: the archive is a function of the five files only


def build_ref(verbose: bool = True) -> str | None:
    """Returns the archive path, or None when neither the reference tree nor a shipped archive exists."""
    if not all(os.path.exists(os.path.join(REF, m)) for m in MEMBERS):
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = {}
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for pkg in PACKAGES:
            z.writestr(zipfile.ZipInfo(pkg + "/__init__.py", _EPOCH), "")
        for m in MEMBERS:
            with open(os.path.join(REF, m), "rb") as f:
                data = f.read()
            manifest[m] = hashlib.sha256(data).hexdigest()
            z.writestr(zipfile.ZipInfo(m, _EPOCH), data)
        z.writestr(zipfile.ZipInfo("MANIFEST.json", _EPOCH),
                   json.dumps(dict(source="showlab/UniVTG (" + REF + ")", members=manifest), indent=1))
    os.replace(tmp, ARCHIVE)
    if verbose:
        print(f"[oracle.build_ref] packed {len(MEMBERS)} reference files -> {os.path.relpath(ARCHIVE, os.path.dirname(HERE))}", file=sys.stderr)
    return ARCHIVE


def _from_archive(mod) -> bool:
    return ARCHIVE in (getattr(mod, "__file__", "") or "")


def read_manifest():
    with zipfile.ZipFile(ARCHIVE) as z:
        return json.loads(z.read("MANIFEST.json"))


def import_ref(names=("model.univtg",)):
    """Import reference modules from the archive (zipimport) and return ([modules], manifest).  Meant for a CHILD process
    (oracle/ref_runner.py): raises ImportError when a foreign `model` / `utils` / `eval` package is already imported, and again when any
    returned module (or its parent package) turns out not to come from the archive -- a shadowed import must not be timed or compared
    silently.  The reference imports scipy.optimize (matcher), numpy, pandas, sklearn, tqdm; all are in the image."""
    if not os.path.exists(ARCHIVE):
        raise ImportError("oracle/_ref/uvtg_reference_model.zip is absent (built by __graft_entry__.build() where /root/reference exists)")
    for name in PACKAGES:
        have = sys.modules.get(name)
        if have is not None and not _from_archive(have):
            raise ImportError(f"a foreign top-level package '{name}' is already imported ({getattr(have, '__file__', '?')})")
    if ARCHIVE not in sys.path:
        sys.path.insert(0, ARCHIVE)
    import importlib
    mods = [importlib.import_module(n) for n in names]
    for n, m in zip(names, mods):
        top = sys.modules[n.split(".")[0]]
        if not (_from_archive(m) and _from_archive(top)):
            raise ImportError(f"'{n}' was not loaded from the archive but from {getattr(m, '__file__', '?')}")
    return mods, read_manifest()


def import_ref_model():
    """(module `model.univtg`, manifest) from the archive; see import_ref."""
    mods, manifest = import_ref(("model.univtg",))
    return mods[0], manifest


if __name__ == "__main__":
    p = build_ref()
    print(p if p else "no reference tree and no shipped archive", file=sys.stderr)
