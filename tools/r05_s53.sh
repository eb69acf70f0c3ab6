set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s53; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_head_gpu.py tests/test_upsample_gpu.py tests/test_aspp_gpu.py tests/test_train_golden.py tests/test_trainer_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
cd /tmp; rm -rf $GRAFT_REPO_ROOT/$O/tr; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o t -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py --modes own --streams main --steps 8 --crop 768 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
grep -E "k_upsample_bwd|k_cosine_bwd|k_dense_bwd_x|k_upsample_fwd|k_cosine_fwd" $(find $O/tr -name "*kernel_stats.csv") | cut -d, -f1-4 | cut -c1-160
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 769 2>&1 | grep -E "^own" | tail -1
