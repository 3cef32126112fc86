#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
OLD=${1:-$R/tools/libuvtg_old.so}
for f in old new; do
  L=$R/univtg_amd/libuvtg.so; [ $f = old ] && L=$OLD
  rm -rf /tmp/p_$f
  UVTG_LIB_PATH=$L rocprofv3 --kernel-trace -d /tmp/p_$f -o $f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-padded-compare > /dev/null 2>&1
  DB=$(find /tmp/p_$f -name '*.db' | head -1)
  python $R/tools/rocpd_seq.py $DB 8 --gemm > $R/gpurun_out/seq_$f.txt
done
paste $R/gpurun_out/seq_old.txt $R/gpurun_out/seq_new.txt | awk '{printf "%-4s %-42s %7s | %-36s %7s\n", $1, $2, $3, $5, $6}'
