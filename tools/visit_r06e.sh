#!/bin/bash
# round-6 visit e: GPU suite, input-pipeline overlap probe, N > 1 step with group A behind layer 0's FFN half, drop-in + pipeline companions re-measured
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/r06e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r06e_pytest_gpu.log | tail -2
grep -n "^FAILED\|^E  " $OUT/r06e_pytest_gpu.log | head -20
UVTG_DEV_ENV=1 UVTG_TN_GROUP_MID0=1 timeout 900 python -m pytest tests -m gpu -q -x -k "overlapped or two_rank" > $OUT/r06e_pytest_mid0.log 2>&1; echo "pytest (group A behind layer 0's FFN half) rc=$?"; grep -n "passed\|failed" $OUT/r06e_pytest_mid0.log | tail -1
timeout 900 python tools/pipeline_overlap_probe2.py > $OUT/r06e_pipeline_overlap_probe.txt 2>&1; tail -14 $OUT/r06e_pipeline_overlap_probe.txt
bash tools/ab5.sh 2 "single-rank step|" "N>1 step on one rank, grouped deferral||--overlap force" "N>1 step, group A behind layer 0's FFN half|UVTG_TN_GROUP_MID0=1|--overlap force" > $OUT/r06e_ab_group_mid0.txt 2>&1; cat $OUT/r06e_ab_group_mid0.txt
( timeout 1200 python bench.py --no-other-configs --no-cpu-baseline 2>$OUT/r06e_bench.err | tail -1 ) > $OUT/r06e_bench_config2.json; cut -c1-200 $OUT/r06e_bench_config2.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06e_bench_config2.json'))
print({k: v for k, v in d['companions']['drop_in_autograd'].items() if k != 'what'})
print({k: v for k, v in d['companions']['with_input_pipeline'].items() if k != 'what'})
PY
