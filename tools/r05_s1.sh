# round-5 session 1: the hygiene changes on the GPU + baseline numbers + the cost of the whole-tile stream-K plan
set -u; cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s1; mkdir -p $O
python -m pytest tests/test_abi_cpu.py tests/test_scorer_gpu.py tests/test_stream_k_coresidency_gpu.py tests/test_pool_scale_gpu.py tests/test_select_gpu.py tests/test_selectors_gpu.py -x -q -m "gpu or not gpu" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for m in on off on off; do echo "MAS_SK_SPLIT=$m"; MAS_SK_SPLIT=$m python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own|ms" | tail -2; done > $O/sk_split_ab.log 2>&1
cat $O/sk_split_ab.log
