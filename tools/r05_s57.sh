set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s57; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_conv_train_gpu.py -q -m gpu -x -k "wgrad or train" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2; do for nb in 1 0; do
  echo "MAS_WGRAD_BX_NB=$nb"; MAS_WGRAD_BX_NB=$nb timeout -k 10 300 python tools/bx_train_table.py --out $O/bx_train_nb${nb}_$rep.md 2>/dev/null | tail -2 | cut -c1-40,150-260
  MAS_WGRAD_BX_NB=$nb timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
done; done
