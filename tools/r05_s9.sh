set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s10; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_bx_gpu.py -x -q -m gpu -k "wgrad_3x3" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 600 python tools/bx_train_table.py --out $O/bx_train_table.md > $O/table.log 2>&1; echo "table rc $?"; awk -F'|' 'NR<=4 || $5 ~ / 3 /' $O/bx_train_table.md
for m in bx f32; do echo "MAS_WGRAD3=$m"; MAS_WGRAD3=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1; done > $O/wgrad3_ab.log 2>&1; cat $O/wgrad3_ab.log
