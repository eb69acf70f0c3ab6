set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s25; mkdir -p $O
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pool --steps 5 --warmup 2 > $O/bench_a.json 2> $O/bench_a.err; echo "bench a rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s25/bench_a.json').read().strip().split('\n')[-1])
print('768 then 769 (empty_cache between):', d['train_iter_ms_768'], d['train_iter_ms_769'], 'pool fwd', d['pool_forward_ms_per_batch'])
PY
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pool --steps 5 --warmup 2 --crop 769 > $O/bench_b.json 2> $O/bench_b.err; echo "bench b rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s25/bench_b.json').read().strip().split('\n')[-1])
print('769 first, 769 again:', d['train_iter_ms_768'], d['train_iter_ms_769'])
PY
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 769 2>&1 | grep -E "^own" | tail -1
