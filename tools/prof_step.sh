#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_step.sh TAG [bench.py args...]
# rocprofv3 kernel trace of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ARGS`, summarised per
# training step into gpurun_out/TAG_kernel_stats.csv (the 47 MB rocpd database is deleted: gpurun_out/
# only travels back when it stays under 64 MiB).
set -e
tag=$1; shift
repo=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --plain "$@" > gpurun_out/${tag}_bench.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" 13 3 gpurun_out/${tag}_kernel_stats.csv > gpurun_out/${tag}_summary.txt
head -3 gpurun_out/${tag}_summary.txt
tail -1 gpurun_out/${tag}_bench.log | cut -c1-300
rm -rf /tmp/prof_$tag
