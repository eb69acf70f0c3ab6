set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s23; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_bn_gpu.py tests/test_train_golden.py tests/test_conv_train_gpu.py -x -q -m gpu -k "epilogue or golden or training or train" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
for m in on off on off; do echo "MAS_BX_STATS=$m"; MAS_BX_STATS=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1; done > $O/bxstats_ab.log 2>&1; cat $O/bxstats_ab.log
MAS_BX_STATS=on timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 769 2>&1 | grep -E "^own" | tail -1
