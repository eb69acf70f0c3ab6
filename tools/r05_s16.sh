set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s16; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bn_gpu.py tests/test_conv_train_gpu.py tests/test_train_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for m in on off on off; do echo "MAS_BN_LASTBLOCK=$m"; MAS_BN_LASTBLOCK=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1; done > $O/bn_ab.log 2>&1; cat $O/bn_ab.log
for m in auto r04; do echo "MAS_TRAIN_BX=$m (769)"; MAS_TRAIN_BX=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 769 2>&1 | grep -E "^own" | tail -1; done
