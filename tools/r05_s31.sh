set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s31; mkdir -p $O
V=$PWD/mulactseg_amd/libvar_pair.so
MAS_LIB=$V timeout -k 10 900 python -m pytest tests/test_conv_bx_gpu.py tests/test_conv_train_gpu.py tests/test_train_golden.py -q -m gpu > $O/pytest_pair.log 2>&1; echo "pytest pair rc $?"; grep -E "FAILED|passed|failed" $O/pytest_pair.log | head -20
for rep in 1 2; do for L in libmulactseg_hip.so libvar_pair.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_table.py --out $O/bx_table_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_train_table.py --out $O/bx_train_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
  echo "$L train:"; MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
done; done > $O/ab.log 2>&1; cat $O/ab.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-train --steps 5 --warmup 2 > $O/bench_pool.json 2> $O/bench_pool.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s31/bench_pool.json').read().strip().split('\n')[-1])
print(json.dumps(d['pool_round']['scan_only'],indent=1)[:1200])
PY
