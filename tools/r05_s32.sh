set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s32; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_conv_train_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "FAILED|passed|failed|Error" $O/pytest.log | head -20
timeout -k 10 300 python tools/bx_table.py --out $O/bx_table.md 2>/dev/null | tail -2
grep -E "\| 3 \| 2 \|" $O/bx_table.md
for m in on off; do echo "MAS_BX_S2K3=$m"; MAS_BX_S2K3=$m timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
MAS_BX_S2K3=$m timeout -k 10 400 python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 3 --warmup 1 --acq-steps 12 > $O/bench_$m.json 2> $O/bench_$m.err; python - <<PY
import json
d=json.loads(open('gpurun_out/s32/bench_$m.json').read().strip().split('\n')[-1])
print('pool forward ms/batch', d.get('pool_forward_ms_per_batch'))
PY
done
timeout -k 10 300 python tools/pool_round_profile.py > $O/round_profile.log 2>&1; head -3 $O/round_profile.log | cut -c1-400; grep -E "pool_valid_mask|click_cost_table|_install_lazy|select_next_batch|calculate_scores|expand_training|synchronize|k4|order|walk" $O/round_profile.log | head -20
