#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 tools/lab/gemm_pp_lab > $OUT/pp_lab.log 2>&1; echo "lab rc=$?"
grep -v "^check.*bad elements 0" $OUT/pp_lab.log
