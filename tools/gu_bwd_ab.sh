# GatedUpdate pair alone (tools/gu_pair_bench.py), in-tree build and the knock-out builds ionic_mpnn_amd/csrc/ab/lib_GUB_*.so
# (-DIMPNN_DIAG_GUB_NOGEMM: the backward's two GEMM passes compiled out; NOCOPY: no compact copies of the rows' inputs)
echo -n "base: "; python tools/gu_pair_bench.py "$@" 2>/dev/null | tail -1
for lib in ionic_mpnn_amd/csrc/ab/lib_GUB_*.so; do
  echo -n "$(basename $lib .so): "; IMPNN_LIB=$PWD/$lib python tools/gu_pair_bench.py "$@" 2>/dev/null | tail -1
done
