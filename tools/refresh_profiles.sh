# (no set -e: a failing diagnostics step must not cost the rest of the evidence)
# One GPU-box run that regenerates the round's evidence under gpurun_out/rN (copy what is to be judged into profiles/).
R=${1:-r3}
mkdir -p gpurun_out/$R
python bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
python bench.py --streams 1 --no-cpu-baseline --no-other-configs > gpurun_out/$R/bench_streams1.json 2>> gpurun_out/$R/bench.err
python bench.py --schedule layered --no-cpu-baseline --no-other-configs > gpurun_out/$R/bench_layered.json 2>> gpurun_out/$R/bench.err
python tools/stamp_profile.py --mode f32t > gpurun_out/$R/stamps.txt 2>&1
python tools/stamp_profile.py --mode f32x3 > gpurun_out/$R/stamps_f32x3.txt 2>&1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$R/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $ROOT/gpurun_out/$R/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$R/prof_s1 -- python3 $ROOT/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-other-configs > $ROOT/gpurun_out/$R/prof_s1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$R/prof_layered -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --schedule layered > $ROOT/gpurun_out/$R/prof_layered.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$R/prof_wide -- python3 $ROOT/tools/wide_probe.py > $ROOT/gpurun_out/$R/prof_wide.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$R/prof_wide_x3 -- python3 $ROOT/tools/wide_probe.py --mode f32x3 > $ROOT/gpurun_out/$R/prof_wide_x3.log 2>&1
cd $ROOT
python tools/wide_probe.py > gpurun_out/$R/wide_probe.json 2>> gpurun_out/$R/bench.err
python tools/wide_probe.py --mode f32x3 >> gpurun_out/$R/wide_probe.json 2>> gpurun_out/$R/bench.err
python tools/wide_mode_bench.py --iters 20 > gpurun_out/$R/wide_modes.json 2>> gpurun_out/$R/bench.err
if [ -f ionic_mpnn_amd/csrc/ab/lib_STAMPS.so ]; then  # (a -DIMPNN_DIAG_WIDE_STAMPS build of the CURRENT sources)
  IMPNN_LIB=$ROOT/ionic_mpnn_amd/csrc/ab/lib_STAMPS.so python tools/wide_stamps.py > gpurun_out/$R/wide_stamps.txt 2>&1 || true
  IMPNN_LIB=$ROOT/ionic_mpnn_amd/csrc/ab/lib_STAMPS.so python tools/wide_stamps.py --mode f32x3 > gpurun_out/$R/wide_stamps_f32x3.txt 2>&1 || true
fi
bash tools/pmc_profile.sh gpurun_out/$R/pmc > /dev/null 2>&1
bash tools/train_profiles.sh gpurun_out/$R/train > /dev/null 2>&1   # training-step timings + config-5 kernel summaries
python tools/gu_pair_bench.py > gpurun_out/$R/gu_pair_bench.jsonl 2>> gpurun_out/$R/bench.err
cat gpurun_out/$R/bench.json
