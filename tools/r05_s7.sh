set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s7; mkdir -p $O
timeout -k 5 60 tools/micro/tr_probe.bin > $O/tr_probe.log 2>&1; cat $O/tr_probe.log
for rep in 1 2; do for L in libmulactseg_hip.so libvar_ilv.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_table.py --out $O/bx_table_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
  echo "$L train:"; MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
done; done > $O/ilv_ab.log 2>&1; cat $O/ilv_ab.log
