#!/bin/bash
# round 4, visit a: fp16 MFMA subnormal probe, whole GPU suite (new fixtures: hl / early-outs / n_input_proj 1,3 / use_txt_pos), bench line in its new form
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
./tools/mfma_f16_probe > $OUT/r04_mfma_f16_subnormal_probe.txt 2>&1; cat $OUT/r04_mfma_f16_subnormal_probe.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|^E  \|Error" $OUT/pytest_gpu.log | head -30
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/smoke.log; cat $OUT/smoke.log
( timeout 600 python bench.py 2>$OUT/bench_err.log | tail -1 ) > $OUT/bench_c2.json; cut -c1-400 $OUT/bench_c2.json; tail -5 $OUT/bench_err.log
