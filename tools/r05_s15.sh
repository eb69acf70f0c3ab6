set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s15; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bn_gpu.py tests/test_train_golden.py tests/test_trainer_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for m in on off on off; do echo "MAS_BN_LASTBLOCK=$m"; MAS_BN_LASTBLOCK=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1; done > $O/bn_ab.log 2>&1; cat $O/bn_ab.log
rm -rf $O/t_tr
rocprofv3 --kernel-trace --output-format csv -d $O/t_tr -o t -- python tools/train_step_probe.py --modes own --streams main --steps 8 --crop 769 > $O/train_probe_769.log 2>&1
python profiles/steady.py $O/t_tr/t_kernel_trace.csv multi_tensor_apply 6 $O/d_train_769_steady.md "rocprofv3 --kernel-trace -- python tools/train_step_probe.py --modes own --streams main --steps 8 --crop 769" > /dev/null
head -40 $O/d_train_769_steady.md
