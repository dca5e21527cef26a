#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_edgeconv_bwd.sh OLD.hip NEW.hip
# A/B of two versions of sug_amd/csrc/edgeconv.hip: each is built into the library in turn (un-profiled hipcc) and
# tools/bench_edgeconv_bwd.py times the backward kernels on the same synthetic smooth features.
set -e
for v in "$1" "$2"; do
  cp "$v" sug_amd/csrc/edgeconv.hip
  make -C sug_amd/csrc -j8 > /dev/null 2>&1
  echo "== $v"
  python3 tools/bench_edgeconv_bwd.py 2>&1 | grep "C="
done
