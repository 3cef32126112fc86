"""Builds univtg_amd/libuvtg.so (HIP kernels + C-ABI, gfx950 only) in-tree with hipcc.

    python -m univtg_amd.build            # incremental
    python -m univtg_amd.build --force

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the source tree.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libuvtg.so")
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "misc.hip", "losses.hip", "postproc.hip", "optim.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value",
         "-Wno-pass-failed"]
# per-source extras.  norm.hip: the row-batched LayerNorm kernels are fully unrolled register tiles; past LLVM's default
# pragma-unroll budget the unroll is silently refused and the tiles become scratch arrays (seen as ScratchSize > 0).
EXTRA_FLAGS = {"norm.hip": ["-mllvm", "-pragma-unroll-threshold=131072"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(os.path.basename(paths[0]), [])).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "uvtg_common.h"), os.path.join(CSRC, "uvtg_kernels.h"),
               os.path.join(os.path.dirname(HERE), "include", "uvtg.h")]
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        st = _stamp([sp] + headers)
        stp = op + ".stamp"
        if force or not os.path.exists(op) or not os.path.exists(stp) or open(stp).read() != st:
            jobs.append((sp, op, st, stp))

    def compile_one(job):
        sp, op, st, stp = job
        r = subprocess.run([hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", op],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {sp}:\n{r.stderr[-4000:]}")
        with open(stp, "w") as f:
            f.write(st)
        return sp
    if jobs:
        if verbose:
            print(f"[univtg_amd.build] compiling {len(jobs)} file(s) for gfx950 ...", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[univtg_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
