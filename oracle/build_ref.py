"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- build-time recipe for `oracle/_ref/` (git-ignored; it travels to the GPU box with the
built tree exactly like `univtg_amd/libuvtg.so` does).

north_star: "the reference's own PyTorch CPU forward is timed on the host cores of the same box in the same run as the reported (not
optimised-against) baseline".  `/root/reference` exists only in the build container, so `__graft_entry__.build()` -- which runs there --
packs the FIVE files of the reference's model path, unmodified and straight from where they lie, into ONE archive

    oracle/_ref/uvtg_reference_model.zip        model/univtg.py  (build_model, Model.forward, SetCriterion: model/univtg.py:105-155,195-351,409-450)
                                                model/transformer_encoder_droppath.py  model/position_encoding.py  model/matcher.py
                                                utils/span_utils.py   (+ empty package markers, + MANIFEST.json with the sha256 of every member)

which `bench.py`'s `cpu_baseline` leg puts on `sys.path` (zipimport) and times (`cpu_baseline.kind = "reference"`); only when the archive is
absent does it fall back to the port (`oracle/nn_baseline.py`, `kind = "port"`) and say so.  No reference source enters the repository: the
archive is a build artefact like the .so, `oracle/_ref/` is listed in .gitignore, and nothing in the product path (`univtg_amd/`) can see it.

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
           "utils/span_utils.py"]
_EPOCH = (1980, 1, 1, 0, 0, 0)          # fixed member timestamps: the archive is a function of the five files only


def build_ref(verbose: bool = True) -> str | None:
    """Returns the archive path, or None when neither the reference tree nor a shipped archive exists."""
    if not all(os.path.exists(os.path.join(REF, m)) for m in MEMBERS):
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = {}
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for pkg in ("model", "utils"):
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


def import_ref_model():
    """Import the reference's model package from the archive (zipimport).  Returns (module `model.univtg`, manifest) or raises ImportError.
    The reference imports `scipy.optimize` (matcher) and `numpy`; both are in the image."""
    if not os.path.exists(ARCHIVE):
        raise ImportError("oracle/_ref/uvtg_reference_model.zip is absent (built by __graft_entry__.build() where /root/reference exists)")
    for name in ("model", "utils"):
        have = sys.modules.get(name)
        if have is not None and ARCHIVE not in (getattr(have, "__file__", "") or ""):
            raise ImportError(f"a foreign top-level package '{name}' is already imported ({getattr(have, '__file__', '?')})")
    if ARCHIVE not in sys.path:
        sys.path.insert(0, ARCHIVE)
    import importlib
    mod = importlib.import_module("model.univtg")
    with zipfile.ZipFile(ARCHIVE) as z:
        manifest = json.loads(z.read("MANIFEST.json"))
    return mod, manifest


if __name__ == "__main__":
    p = build_ref()
    print(p if p else "no reference tree and no shipped archive", file=sys.stderr)
