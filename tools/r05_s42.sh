set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s42; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_train_golden.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "FAILED|passed|failed" $O/pytest.log | head
cd /tmp; rm -rf $GRAFT_REPO_ROOT/$O/tr; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o t -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py --modes own --streams main --steps 8 --crop 768 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
grep -E "k_bx_pack_multi|k_sk_pack_multi" $(find $O/tr -name "*kernel_stats.csv") | cut -c1-200
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
