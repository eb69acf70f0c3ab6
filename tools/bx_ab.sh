# A/B of library builds on the per-layer table of the inference convolutions in ONE GPU session:
#   bash tools/bx_ab.sh libmulactseg_hip.so libvar_x.so ...     (names under build/variants/, or "product")
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = product ]; then unset MAS_LIB; else export MAS_LIB=$PWD/build/variants/$L; fi
    python tools/bx_table.py --out gpurun_out/bx_table_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
  done
done
