"""Dev tool: A/B of two builds of libuvtg.so at kernel level (same process order, interleaved rounds via subprocesses).
    python tools/nt_ab.py [old.so]      # runs itself once per library with UVTG_LIB_PATH set"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from univtg_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    lib.uvtg_debug_force_nt_tile(256)
    for (M, N, K) in [(24300, 1024, 1024), (27392, 1024, 1024), (24300, 3072, 1024), (24300, 1024, 3072)]:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = []
        for bm in (256, 192, 128):
            lib.uvtg_debug_force_nt_bm(bm)
            for act in (100, 0):
                t = min(timeit(lambda: ops.linear_bf16(a, w, None, act)) for _ in range(2))
                out.append(f"bm{bm}/{'loop' if act == 100 else 'f32out'} {t:6.1f}us {2*M*N*K/t/1e6:5.0f}TF")
        print(f"{M}x{N}x{K}: " + " | ".join(out))
    for (M, N, K) in [(24300, 1024, 1024), (24300, 2048, 1024), (19200, 1024, 3072)]:
        dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        nf = lib.uvtg_wgrad_scratch_floats(M, N, K)
        t = timeit(lambda: ops.wgrad_bf16_ws(dy, x))
        print(f"TN256 {M}x{N}x{K}: {t:7.1f} us {2*M*N*K/t/1e6:6.0f} TF (incl. zero-init + scratch alloc)")
    sys.exit(0)
old = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "libuvtg_old.so")
for name, path in (("OLD", old), ("NEW", os.path.join(ROOT, "univtg_amd", "libuvtg.so")), ("OLD", old), ("NEW", os.path.join(ROOT, "univtg_amd", "libuvtg.so"))):
    print(f"==== {name} {path}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, UVTG_LIB_PATH=path), check=False)
