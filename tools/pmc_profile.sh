#!/bin/bash
# Collects PMC counters for the bench's dominant kernel in separate passes (rocprofv3 --pmc only,
# never combined with traces), one stream so that every dispatch has the chip to itself.
# Run on the GPU box from the repo root:  bash tools/pmc_profile.sh <outdir>   (PMC_BENCH_ARGS: extra bench.py flags)
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python3 "$ROOT/bench.py" --steps 5 --warmup 2 --ramp-ms 20 --streams 1 --no-cpu-baseline --no-other-configs ${PMC_BENCH_ARGS:-} > "$ROOT/$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MUL_F32
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
