set -u; cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_bx_gpu.py -x -q -m gpu -k "split or non_finite or integers" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 500 python tools/bx_splitk_sweep.py --out $O/bx_splitk.md > $O/sweep.log 2>&1; echo "sweep rc $?"; tail -3 $O/sweep.log
for m in auto r04 auto r04; do echo "MAS_TRAIN_BX=$m"; MAS_TRAIN_BX=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1; done > $O/train_ab.log 2>&1
cat $O/train_ab.log
