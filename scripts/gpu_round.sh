#!/bin/bash
# One GPU-box session: parity tests, bench, kernel-trace profile.  Stops after any step that times out.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/round.log
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/round.log; exit 1; fi
  return 0
}
: > gpurun_out/round.log
step pytest_gpu 700 python -m pytest tests -m gpu -q -x
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
step bench 400 python bench.py --mixed
PROF=gpurun_out/prof_bench
rm -rf $PROF
step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $PROF -- python3 bench.py --steps 5 --warmup 1 --gibbs-sweeps 10 --no-cpu
find $PROF -name "*kernel_stats.csv" | head -3 | while read f; do echo "--- $f"; head -30 "$f"; done | tee gpurun_out/kernel_stats_head.txt
exit 0
