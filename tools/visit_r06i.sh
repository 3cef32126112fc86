#!/bin/bash
# round 6, visit i/j: GPU suite + per-kernel table of the step + step A/B (env switch or previous library tools/libuvtg_prev.so)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/visit_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/visit_pytest.log | tail -2 | cut -c1-200; grep -n "FAILED\|^E  " $OUT/visit_pytest.log | head
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-companions --no-other-configs"
UVTG_DEV_ENV=1 bash tools/prof.sh step_new 20 $BENCH > /dev/null 2>&1
grep "ln_dgb\|wide_wave\|heads_saliency_fwd" $OUT/step_new_stats.md | cut -d'|' -f2-7 | cut -c1-200
grep "gemm_nt256" $OUT/step_new_stats.md | cut -d'|' -f2-7 | cut -c1-200 > $OUT/nt_new.txt
UVTG_DEV_ENV=1 UVTG_PROJ_DGRAD_F32=1 bash tools/prof.sh step_old 20 $BENCH > /dev/null 2>&1
grep "ln_dgb" $OUT/step_old_stats.md | cut -d'|' -f2-7 | cut -c1-200
grep "gemm_nt256" $OUT/step_old_stats.md | cut -d'|' -f2-7 | cut -c1-200 > $OUT/nt_old.txt
diff $OUT/nt_new.txt $OUT/nt_old.txt | head -20
bash tools/ab5.sh 4 "bf16 dgrad stream|" "fp32 (rounds 1-5)|UVTG_PROJ_DGRAD_F32=1" 2>&1 | tee $OUT/ab_proj_dgrad_bf16.txt
