# rocprofv3 kernel summary of the training step (tools/train_bench.py); run on the GPU box from the repo root.
# usage: bash tools/profile_train.sh <batch> [extra train_bench flags]
set -e
B=${1:-32}; shift || true
ROOT=$PWD
OUT=$ROOT/gpurun_out/r1/prof_train_b$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/train_bench.py --batch $B "$@" > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | sort | sed -n 1p)
if [ -n "$f" ]; then cut -d, -f1-4,7 "$f" | sed -n 1,40p; else echo "no kernel_stats.csv under $OUT"; fi
