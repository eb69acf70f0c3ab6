set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s59; mkdir -p $O gpurun_out/p5
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout -k 10 900 bash tools/profile_r05.sh train > gpurun_out/p5/train.log 2>&1; echo "train rc $?"; sed -n 5,5p gpurun_out/p5/c_train_768_steady.md
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s59/bench.json').read().strip().split('\n')[-1])
for k in ('value','train_iter_ms_768','train_iter_ms_769','pool_forward_ms_per_batch','loss_gpu_ms_fwd_bwd','pool_round_scan_only_s','pool_round_with_model_s'):
    print(k, d.get(k))
print('roofline frac', d['roofline']['frac'])
PY
