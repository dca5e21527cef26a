#!/bin/bash
# usage (GPU box, repo root): tools/prof_cmd.sh TAG script.py [args...]
# rocprofv3 kernel trace of `python3 script.py args`, aggregated over the whole run into
# gpurun_out/TAG_kernel_stats.csv (top kernels by total time); the rocpd database is deleted.
set -e
tag=$1; shift
repo=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python3 "$@" > gpurun_out/${tag}_run.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" 1 0 gpurun_out/${tag}_kernel_stats.csv > gpurun_out/${tag}_summary.txt
head -${PROF_LINES:-25} gpurun_out/${tag}_summary.txt | cut -c1-160
rm -rf /tmp/prof_$tag
