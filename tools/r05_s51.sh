set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s51; mkdir -p $O
V=$PWD/mulactseg_amd/libvar_wxp3.so
MAS_LIB=$V timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_conv_train_gpu.py -q -m gpu -x -k "wgrad or train" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2; do for L in libmulactseg_hip.so libvar_wxp3.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_train_table.py --out $O/bx_train_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /" | cut -c1-40,150-260
  echo "$L train:"; MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
done; done
