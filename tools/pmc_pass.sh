#!/bin/bash
# usage (GPU box, repo root): tools/pmc_pass.sh TAG script.py args...
# Two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass; counters are collected
# with --kernel-trace only, as the pool requires) of `python3 script.py args`; per-kernel averages in
# gpurun_out/TAG_pmc.txt (KB per dispatch; on gfx950 FETCH_SIZE counts wide coalesced reads at half
# their bytes, MI355X_MICROARCH.md).
set -e
tag=$1; shift
repo=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$repo"
: > gpurun_out/${tag}_pmc.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$tag
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 "$@" > gpurun_out/${tag}_pmc_run.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr >> gpurun_out/${tag}_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get('Counter_Name') == sys.argv[2]:
        agg[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print('%-10s %-72s dispatches %3d  avg %12.1f  last %12.1f' % (sys.argv[2], k, len(v), sum(v) / len(v), v[-1]))
PY
done
cat gpurun_out/${tag}_pmc.txt
rm -rf /tmp/pmc_$tag
