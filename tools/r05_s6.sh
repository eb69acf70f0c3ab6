set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s6; mkdir -p $O
for rep in 1 2; do for L in libmulactseg_hip.so libvar_pipe.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_table.py --out $O/bx_table_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
  echo "$L train:"; MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
done; done > $O/pipe_ab.log 2>&1; cat $O/pipe_ab.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; tail -5 $O/pytest_full.log
