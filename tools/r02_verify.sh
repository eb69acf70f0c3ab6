# round-2 verification pass: GPU tests, the driver's own bench command, the default bench, kernel stats + PMC traffic
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests -m gpu -x -q > gpurun_out/r2b_gputests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2b_gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2b_bench_driver.json 2> gpurun_out/r2b_bench_driver.err; echo "driver-style bench rc=$?"
CMD="python bench.py --no-cpu-baseline --no-train --no-pool"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b_stats -o f -- $CMD > gpurun_out/r2b_stats_bench.json 2>/dev/null; echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2b_fetch -o f -- $CMD --steps 20 --warmup 2 --ramp 0 > /dev/null 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2b_write -o f -- $CMD --steps 20 --warmup 2 --ramp 0 > /dev/null 2>&1; echo "write rc=$?"
find gpurun_out/r2b_stats gpurun_out/r2b_fetch gpurun_out/r2b_write -name "*.csv" | head -20
