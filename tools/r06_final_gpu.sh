# round-6 closing GPU session: probe table, stream bit checks, full GPU suite, the driver's bench command
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python tools/pk_opsel_probe.py > gpurun_out/r06_pk_probe.log 2>&1 && echo "probe done"
python tools/stream_bits_check.py --crop 768 > gpurun_out/r06_stream_bits_768.log 2>&1; tail -1 gpurun_out/r06_stream_bits_768.log
python tools/stream_bits_check.py --crop 769 --reps 4 > gpurun_out/r06_stream_bits_769.log 2>&1; tail -1 gpurun_out/r06_stream_bits_769.log
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_suite.log 2>&1; tail -2 gpurun_out/r06_gpu_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench6.json 2> gpurun_out/r06_bench6.err; tail -c 600 gpurun_out/r06_bench6.json
