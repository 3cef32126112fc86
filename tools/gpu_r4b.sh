#!/bin/bash
# round 4, visit b: split-operand (fp16 hi/lo) precise path + n_input_proj / use_txt_pos engine: kernel tests, model tests, full-size parity,
# tiny_hl gradient diagnosis, bench + inference lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error\|^\[" $OUT/pytest_gpu.log | cut -c1-300 | head -60
timeout 120 python tools/diag_hl.py > $OUT/diag_hl.log 2>&1; tail -40 $OUT/diag_hl.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/smoke.log; cat $OUT/smoke.log
( timeout 600 python bench.py --no-cpu-baseline 2>$OUT/bench_err.log | tail -1 ) > $OUT/bench_c2.json; cut -c1-300 $OUT/bench_c2.json; tail -3 $OUT/bench_err.log
( timeout 400 python bench.py --mode infer 2>$OUT/infer_err.log | tail -1 ) > $OUT/bench_infer.json; cut -c1-200 $OUT/bench_infer.json; tail -3 $OUT/infer_err.log
