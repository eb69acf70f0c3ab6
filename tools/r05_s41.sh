set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s41; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_aspp_gpu.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "FAILED|passed|failed" $O/pytest.log | head
