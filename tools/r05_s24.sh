set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/p5 gpurun_out/s24
timeout -k 10 900 bash tools/profile_r05.sh train > gpurun_out/p5/train.log 2>&1; echo "train rc $?"; head -12 gpurun_out/p5/c_train_768_steady.md
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s24/bench.json 2> gpurun_out/s24/bench.err; echo "bench rc $?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s24/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/s24/smoke.log
