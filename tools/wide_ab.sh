# rocprofv3 kernel stats of tools/wide_probe.py for every diagnostics build under ionic_mpnn_amd/csrc/ab/lib_*.so
# (usage on the GPU box: bash tools/wide_ab.sh <outdir> [probe args])
ROOT=$PWD
OUT=$ROOT/${1:-gpurun_out/wide_ab}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/ionic_mpnn_amd/csrc/ab/lib_*.so; do
  v=$(basename $lib .so)
  export IMPNN_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $ROOT/tools/wide_probe.py "$@" > $OUT/$v.log 2>&1
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $(tail -1 $OUT/$v.log)"
  python3 $ROOT/tools/kstats.py $f 1 6 | grep -E "wide_|total"
done
