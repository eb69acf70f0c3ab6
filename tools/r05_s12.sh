set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s13; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_bx_gpu.py -x -q -m gpu -k "wgrad_3x3" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout -k 10 600 python tools/bx_train_table.py --out $O/bx_train_table.md > $O/table.log 2>&1; echo "table rc $?"; awk -F'|' 'NR<=4 || $5 ~ / 3 /' $O/bx_train_table.md
bash tools/pmc_bx_kernels.sh 2>&1 | grep "k_wgrad_bx3" | head -2
