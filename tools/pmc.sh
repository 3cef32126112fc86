#!/bin/bash
# (every pass under `timeout`: a counter set the hardware cannot collect aborts rocprofv3, which then hangs in its signal handler -- one such
#  pass cost a 15-minute visit)
# usage: tools/pmc.sh <name> "<counters pass 1>" ["<counters pass 2>" ...] -- <command...>   (one rocprofv3 --pmc run per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}; N=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done; shift
cd /tmp; export TMPDIR=/tmp
i=0
for P in "${PASSES[@]}"; do
  rm -rf /tmp/pmc_$N_$i
  timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --pmc $P $PMC_EXTRA --output-format csv -d /tmp/pmc_${N}_$i -o p -- "$@" > $R/gpurun_out/pmc_${N}_$i.log 2>&1
  F=$(find /tmp/pmc_${N}_$i -name '*counter_collection.csv' | head -1)
  python - "$F" <<'PY' | tee $R/gpurun_out/pmc_${N}_$i.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
  i=$((i+1))
done
