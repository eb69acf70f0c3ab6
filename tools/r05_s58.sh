set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s58; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_conv_train_gpu.py tests/test_train_golden.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 769 2>&1 | grep -E "^own" | tail -1
