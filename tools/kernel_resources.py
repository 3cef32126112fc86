"""Dev tool: registers / scratch / LDS / reachable waves per SIMD of every kernel in the built libuvtg.so (code-object metadata)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univtg_amd import build
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k in build.kernel_resources():
    n = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if flt in n:
        print(f'{n[:70]:70s} regs {k["vgpr"]:3d} (acc {k["agpr"]:3d}) scratch {k["scratch"]:5d} lds {k["lds"]:6d} wg {k["max_flat_workgroup_size"]:4d} waves/simd {k["waves_per_simd"]}')
