#!/bin/bash
# local wrapper around gpurun: stamps the commit the snapshot is taken from (the GPU box has no .git), then runs the visit
#   tools/gpu.sh <timeout-seconds> '<command>'
cd /root/repo
( git rev-parse --short HEAD; git status --porcelain | grep -v '^??' | head -1 | sed 's/.*/+dirty/' ) | tr -d '\n' > tools/.git_head
T=$1; shift
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
