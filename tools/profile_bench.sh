set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python bench.py > gpurun_out/bench_r1f.json 2> gpurun_out/bench_r1f.err; tail -c 600 gpurun_out/bench_r1f.err
CMD="python bench.py --no-cpu-baseline --no-train"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f_stats -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/f_fetch -o f -- $CMD --steps 20 --warmup 2 --ramp 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/f_write -o f -- $CMD --steps 20 --warmup 2 --ramp 0 > /dev/null 2>&1
find gpurun_out/f_stats gpurun_out/f_fetch gpurun_out/f_write -name "*.csv" | head
