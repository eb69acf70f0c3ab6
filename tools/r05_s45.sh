set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/p5 gpurun_out/s45
timeout -k 10 900 bash tools/profile_r05.sh train > gpurun_out/p5/train.log 2>&1; echo "train rc $?"; head -10 gpurun_out/p5/c_train_768_steady.md | tail -5
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s45/bench.json 2> gpurun_out/s45/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s45/bench.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','train_iter_ms_768','train_iter_ms_769','pool_forward_ms_per_batch','loss_gpu_ms_fwd_bwd','pool_round_scan_only_s','pool_round_with_model_s'):
    print(k, d.get(k))
print('roofline frac', d['roofline']['frac'])
PY
