#!/bin/bash
# PMC counters of the config-5 training step's kernels (atom_dim 128, 6 steps, batch 4096, eager, ONE stream so that the
# per-dispatch counters belong to one kernel at a time) in separate rocprofv3 --pmc passes (never combined with traces).
# Run on the GPU box from the repo root:  bash tools/pmc_profile_train.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_train}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export IMPNN_TWO_STREAM_MAX_BATCH=0
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python3 "$ROOT/tools/train_bench.py" --batch 4096 --atom-dim 128 --steps 6 --iters 2 > "$ROOT/$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE GRBM_GUI_ACTIVE
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
