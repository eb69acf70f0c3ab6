set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s30; mkdir -p $O
V=$PWD/mulactseg_amd/libvar_sh16.so
MAS_LIB=$V timeout -k 10 900 python -m pytest tests/test_conv_train_gpu.py tests/test_bn_gpu.py tests/test_train_golden.py -q -m gpu > $O/pytest_sh16.log 2>&1; echo "pytest sh16 rc $?"; grep -E "FAILED|passed|failed" $O/pytest_sh16.log | head -40
timeout -k 10 300 python tools/pool_round_profile.py > $O/round_profile.log 2>&1; head -70 $O/round_profile.log
