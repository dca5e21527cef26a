#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_edgeconv_writes.sh
# FETCH_SIZE / WRITE_SIZE (KB per dispatch) of the fused EdgeConv layer's two kernels at the layer-2 shape
# (64 clouds x 1024 points, C = 64 -> Co = 64, no backward), launched through the C ABI on buffers that are reused from
# launch to launch (tools/bench_edgeconv_fused.py EF_ONLY): no other kernel's dirty L2 lines are written back inside
# these dispatches.  Variants: product | main kernel alone | without the arg store | without the z store.
cd /tmp && export TMPDIR=/tmp
for v in " " "-DSUG_EF_ABL_NOACT" "-DSUG_EF_ABL_NOACT -DSUG_EF_ABL_NOARG" "-DSUG_EF_ABL_NOACT -DSUG_EF_ABL_NOZ"; do
  # step 1, NOT profiled: hipcc builds the variant (hipcc execs clang/lld -- never under the profiler's preload)
  python3 $GRAFT_REPO_ROOT/tools/bench_edgeconv_fused.py --build-only "$v" || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/px
    # step 2, profiled: the process only ctypes-loads the prebuilt library (no child processes); one counter per pass
    EF_ONLY="$v" rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/px -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_edgeconv_fused.py > /tmp/px.log 2>&1 || { tail -20 /tmp/px.log; exit 1; }
    f=$(find /tmp/px -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$ctr" "$v" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == sys.argv[2] and 'edgeconv' in r['Kernel_Name']:
        agg[r['Kernel_Name'].split('(')[0][-60:]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('[%s] %-10s %-60s dispatches %d median %10.1f KB' % (sys.argv[3].strip() or 'product', sys.argv[2], k, len(v), sorted(v)[len(v) // 2]))
PY
  done
done
