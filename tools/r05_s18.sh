set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s18; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_conv_train_gpu.py -x -q -m gpu -k "split or integers or train" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 500 python tools/bx_splitk_sweep.py --shape train769 --out $O/bx_splitk_769.md > $O/sweep.log 2>&1; echo "sweep rc $?"; grep " 3 | " $O/bx_splitk_769.md | cut -c1-260
for c in 769 768; do timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop $c 2>&1 | grep -E "^own" | tail -1; done
