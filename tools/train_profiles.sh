# Training-step figures of a round: tools/train_bench.py at the five shapes of profiles/r2_train_bench.jsonl, and the
# rocprofv3 kernel summaries of the config-5 step (atom_dim 128, 6 steps) at batch 4096 (eager) and 32 (graphed).
# usage (GPU box, repo root): bash tools/train_profiles.sh <out dir under gpurun_out/>
OUT=${1:-gpurun_out/train_profiles}
mkdir -p $OUT
: > $OUT/train_bench.jsonl
timeout -k 10 200 python tools/train_bench.py --batch 32 --graph >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
timeout -k 10 200 python tools/train_bench.py --batch 4096 --graph >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
timeout -k 10 200 python tools/train_bench.py --batch 32 --graph --atom-dim 128 --steps 6 --iters 50 >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
timeout -k 10 200 python tools/train_bench.py --batch 256 --graph --atom-dim 128 --steps 6 --iters 50 >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
timeout -k 10 200 python tools/train_bench.py --batch 4096 --atom-dim 128 --steps 6 >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
timeout -k 10 200 python tools/train_bench.py --batch 4096 --atom-dim 128 --steps 6 --graph >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
timeout -k 10 200 python tools/train_bench.py --batch 256 --atom-dim 128 --steps 6 --graph --explicit-h >> $OUT/train_bench.jsonl 2>> $OUT/err.log &&
bash tools/profile_train.sh 4096 --atom-dim 128 --steps 6 > $OUT/prof4096.txt 2>&1 &&
cp $(find gpurun_out/r1/prof_train_b4096 -name "*kernel_stats.csv" | sort | sed -n 1p) $OUT/train_config5_b4096_kernel_stats.csv &&
bash tools/profile_train.sh 32 --graph --atom-dim 128 --steps 6 > $OUT/prof32.txt 2>&1 &&
cp $(find gpurun_out/r1/prof_train_b32 -name "*kernel_stats.csv" | sort | sed -n 1p) $OUT/train_config5_b32_kernel_stats.csv
