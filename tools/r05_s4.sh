set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_losses_gpu.py tests/test_train_golden.py -x -q -m gpu -k "presplit or fused or golden or training_mode" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout -k 10 600 python tools/bx_pre_table.py --shape pool --out $O/bx_pre_pool.md > $O/pre_pool.log 2>&1; echo "pre pool rc $?"; tail -4 $O/pre_pool.log
timeout -k 10 300 python tools/bx_pre_table.py --shape train --out $O/bx_pre_train.md > $O/pre_train.log 2>&1; echo "pre train rc $?"; tail -3 $O/pre_train.log
timeout -k 10 300 python tools/loss_probe.py > $O/loss_probe.log 2>&1; cat $O/loss_probe.log
