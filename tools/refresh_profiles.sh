set -e
mkdir -p gpurun_out/r1
python bench.py > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err
python bench.py --mode f32 --no-cpu-baseline > gpurun_out/r1/bench_f32.json 2>> gpurun_out/r1/bench.err
python bench.py --schedule layered --no-cpu-baseline > gpurun_out/r1/bench_layered.json 2>> gpurun_out/r1/bench.err
python tools/stamp_profile.py > gpurun_out/r1/stamps.txt 2>&1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r1/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/r1/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r1/prof_layered -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --schedule layered > $ROOT/gpurun_out/r1/prof_layered.log 2>&1
cd $ROOT
bash tools/pmc_profile.sh gpurun_out/r1/pmc > /dev/null 2>&1
cat gpurun_out/r1/bench.json
