set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_losses_gpu.py tests/test_trainer_gpu.py tests/test_train_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 300 python tools/loss_probe.py --stepwise > $O/loss_probe.log 2>&1; cat $O/loss_probe.log
timeout -k 10 300 python tools/loss_probe.py --crop 769 >> $O/loss_probe.log 2>&1
rm -rf $O/l_stats $O/l_fetch $O/l_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l_stats -o l -- python tools/loss_probe.py --iters 30 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/l_fetch -o l -- python tools/loss_probe.py --iters 10 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/l_write -o l -- python tools/loss_probe.py --iters 10 > /dev/null 2>&1
python profiles/summarize.py stats $(find $O/l_stats -name "*kernel_stats.csv") $O/n_loss_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/loss_probe.py --iters 30" && head -30 $O/n_loss_kernel_stats.md
python profiles/pmc_kernels.py $(find $O/l_fetch -name "*counter_collection.csv") $(find $O/l_write -name "*counter_collection.csv") $O/n_loss_hbm_pmc.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/loss_probe.py --iters 10" && head -30 $O/n_loss_hbm_pmc.md
for m in on off; do echo "MAS_SK_SPLIT=$m"; MAS_SK_SPLIT=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1; done > $O/sk_split_ab.log 2>&1; cat $O/sk_split_ab.log
