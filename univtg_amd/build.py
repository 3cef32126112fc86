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
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "misc.hip", "losses.hip", "postproc.hip", "detr.hip", "optim.hip", "engine.hip"]
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
               os.path.join(os.path.dirname(HERE), "include", "uvtg.h"), os.path.join(os.path.dirname(HERE), "include", "uvtg_dev.h")]
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


def kernel_resources(lib: str = LIB):
    """Per-kernel register / scratch / LDS use read from the gfx950 code objects embedded in ``libuvtg.so`` (AMDGPU metadata
    notes).  rocprofv3's ``vgpr`` column does not show the accumulator half or what hipcc really allocated; this does:
    ``waves_per_simd = 512 // (vgpr_count rounded to 8)`` is the occupancy the kernel can reach (before LDS limits).
    Returns a list of dicts: name, vgpr (unified total), agpr (the accumulation part of it), sgpr, scratch, lds, max_flat_workgroup_size, waves_per_simd."""
    import re
    import shutil
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    tmp = tempfile.mkdtemp(prefix="uvtg_co_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], cwd=tmp, capture_output=True, text=True, check=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
                name = g("name")
                if not name:
                    continue
                agpr = int(re.match(r"\s*(\d+)", blk).group(1))
                vgpr = int(g("vgpr_count").group(1))
                tot = (vgpr + 7) // 8 * 8            # .vgpr_count is the unified total (architectural + accumulation registers)
                out.append(dict(name=name.group(1), vgpr=vgpr, agpr=agpr, sgpr=int(g("sgpr_count").group(1)),
                                scratch=int(g("private_segment_fixed_size").group(1)), lds=int(g("group_segment_fixed_size").group(1)),
                                max_flat_workgroup_size=int(g("max_flat_workgroup_size").group(1)),
                                waves_per_simd=min(8, 512 // max(tot, 8))))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
