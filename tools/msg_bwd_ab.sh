# message backward alone, in-tree build and every diagnostics build under ionic_mpnn_amd/csrc/ab/lib_NO*.so
# (usage on the GPU box: bash tools/msg_bwd_ab.sh)
for args in "--batch 4096" "--batch 256 --explicit-h" "--batch 32"; do
  echo "== $args"
  echo -n "base: "; python tools/msg_bwd_bench.py $args 2>/dev/null | tail -1
  for lib in ionic_mpnn_amd/csrc/ab/lib_NO*.so; do
    echo -n "$(basename $lib .so): "; IMPNN_LIB=$PWD/$lib python tools/msg_bwd_bench.py $args 2>/dev/null | tail -1
  done
done
