set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s21; mkdir -p $O
for rep in 1 2; do for L in libmulactseg_hip.so libvar_epad.so; do
  echo "$L:"; MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 400 python tools/bx_train_table.py --out $O/table_${L%.so}_$rep.md 2>/dev/null | tail -1
done; done > $O/epad_ab.log 2>&1; cat $O/epad_ab.log
