#!/bin/bash
# tools/libuvtg_trace.so: the library with gemm.hip rebuilt under -DUVTG_NT_TRACE (per-tile phase timestamps, tools/nt_trace.py)
set -e
R=$(cd $(dirname $0)/.. && pwd)
python -c "import sys; sys.path.insert(0, '$R'); from univtg_amd import build; build.build()"
mkdir -p /tmp/uvtg_trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-value -Wno-pass-failed -DUVTG_NT_TRACE \
  -c $R/univtg_amd/csrc/gemm.hip -o /tmp/uvtg_trace/gemm.o
ls $R/univtg_amd/csrc/build/*.o | grep -v '/gemm.o' | xargs /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/libuvtg_trace.so /tmp/uvtg_trace/gemm.o
