#!/bin/bash
# round-6 visit b: the repaired tests, the epilogue-overlap probe, A/B of the part-ordered hybrid fold against the previous build is implicit (same box: headline twice)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
rm -f $OUT/index_clause.txt
timeout 1500 python -m pytest tests -m gpu -q -x -k "loader_waves or backward_reads or fused_adamw or real_reference or post_nms_indices or config2_full_size_fp32x3 or conv_head_weight or native_train_step or two_rank" > $OUT/r06b_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06b_pytest.log | cut -c1-300
grep -n "FAILED\|^E  " $OUT/r06b_pytest.log | head -20
cat $OUT/index_clause.txt
timeout 600 python tools/epilogue_overlap_probe.py > $OUT/r06b_epilogue_overlap_probe.txt 2>&1; cat $OUT/r06b_epilogue_overlap_probe.txt | tail -8
bash tools/ab5.sh 2 "default|" > $OUT/r06b_headline.txt 2>&1; cat $OUT/r06b_headline.txt
