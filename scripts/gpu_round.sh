#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel-trace profile, PMC passes.
# Stops after any step that times out.  Usage: bash scripts/gpu_round.sh [tag] [a|b|ab]   (a: tests, smoke, bench,
# kernel trace; b: the PMC passes -- a gpurun call is capped at 20 minutes, the two halves fit one call each)
set -o pipefail
TAG=${1:-r03}
PHASE=${2:-ab}
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/round.log
  tail -n 4 "gpurun_out/$name.log" | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/round.log; exit 1; fi
  return 0
}
: > gpurun_out/round.log
if [[ $PHASE == *a* ]]; then
step pytest_gpu 600 python -m pytest tests -m gpu -q
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py
BARGS="--steps 5 --warmup 1 --gibbs-sweeps 10 --gibbs-chain 0 --no-cpu"
rm -rf gpurun_out/prof_$TAG
step rocprof 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py $BARGS
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" gpurun_out/kernel_stats_$TAG.csv; done
grep -h "^{" gpurun_out/bench.log | tail -1 > gpurun_out/bench_line_$TAG.json
fi
if [[ $PHASE == *b* ]]; then
rm -rf gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_valu_$TAG gpurun_out/pmc_stall_$TAG gpurun_out/pmc_mfma_$TAG gpurun_out/pmc_lanes_$TAG
PARGS="--steps 2 --warmup 1 --gibbs-sweeps 3 --gibbs-chain 0 --no-cpu --c5-sweeps 2 --mlogit-n 0"
step pmc_fetch 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 bench.py $PARGS
step pmc_write 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 bench.py $PARGS
step pmc_valu 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_valu_$TAG -- python3 bench.py $PARGS
step pmc_stall 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_stall_$TAG -- python3 bench.py $PARGS
step pmc_mfma 400 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_$TAG -- python3 bench.py $PARGS
step pmc_lanes 400 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_lanes_$TAG -- python3 bench.py $PARGS
python3 scripts/summarize_pmc.py $TAG > gpurun_out/pmc_summary_$TAG.txt 2>&1
cat gpurun_out/pmc_summary_$TAG.txt | cut -c1-300
fi
exit 0
