"""Dev tool (round 6): where does the exposed epilogue time of the persistent NT GEMM go?  Kernel-level entry (uvtg_linear_bf16: general
epilogue, fp32 output, no epilogue operand) on tools/libuvtg_nostore.so (tools/build_nostore.py) at the headline's launch shapes:
    act 100  main loop only            act 104  main loop + the whole epilogue WITHOUT its output stores            act 0  the full launch
    act 105  the full launch WITHOUT the epilogue's LDS transpose round trip (garbage values, same loads / arithmetic / stores)
The measurement build must agree with the shipped one in act 0 (checked here against a float64 product), and its act-0 time is printed beside
the shipped library's so that a broken build cannot pass for a fast one.
    python tools/nt_epilogue_parts.py          # runs itself once per library"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from univtg_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    def timeit(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    lib.uvtg_debug_force_nt_tile(256)
    modes = (100, 104, 105, 0) if sys.argv[2] == "measure" else (100, 0)
    for (M, N, K) in [(27392, 1024, 1024), (27392, 3072, 1024), (27392, 1024, 3072)]:
        g = torch.Generator().manual_seed(M + N + K)
        a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(torch.bfloat16)
        got = ops.linear_bf16(a, w, None, 0)
        ref = a[:512].double() @ w.double().t()
        err = float((got[:512].double() - ref).abs().max() / ref.abs().max())
        assert bool(torch.isfinite(got).all()) and err < 2e-5, (M, N, K, err)
        for bm in (256,):
            lib.uvtg_debug_force_nt_bm(bm)
            t = {m: min(timeit(lambda: ops.linear_bf16(a, w, None, m)) for _ in range(3)) for m in modes}
            line = f"{M} x {N} x {K}, {bm}-row tiles: loop only {t[100]:6.1f} us"
            if 104 in t:
                line += f" | + epilogue without stores {t[104]:6.1f} us (+{t[104] - t[100]:4.1f}) | full without the LDS transpose {t[105]:6.1f} us ({t[105] - t[0]:+5.1f} vs full)"
            line += f" | full (fp32 out, {M * N * 4 / 1e6:.0f} MB) {t[0]:6.1f} us (+{t[0] - t[100]:4.1f} over the loop" + (f", stores {t[0] - t[104]:4.1f})" if 104 in t else ")")
            print(line + f"   [act 0 vs float64: {err:.1e}]", flush=True)
    lib.uvtg_debug_force_nt_bm(0); lib.uvtg_debug_force_nt_tile(0)
    sys.exit(0)
for rnd in range(2):
    for name, path, kind in (("shipped libuvtg.so", os.path.join(ROOT, "univtg_amd", "libuvtg.so"), "ship"), ("measurement build", os.path.join(ROOT, "tools", "libuvtg_nostore.so"), "measure")):
        print(f"==== {name}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind], env=dict(os.environ, UVTG_LIB_PATH=path), check=False)
