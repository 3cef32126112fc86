#!/bin/bash
# round 4, visit f: whole GPU suite on the final build, smoke, then the artifact set again (final kernels: PMC hash must match) + variant-B and config-4 step tables
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|^FAILED\|^\[config2 fp32\|^\[seeds\|   sample" $OUT/pytest_gpu.log | cut -c1-220 | head -40
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > $OUT/smoke.log; cat $OUT/smoke.log
bash tools/artifacts_r04.sh r04
timeout 300 bash tools/prof.sh r04c2B 26 python $R/bench.py --variant B --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
timeout 300 bash tools/prof.sh r04c4 26 python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-companions > /dev/null 2>&1
