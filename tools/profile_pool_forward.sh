# kernel trace of the acquisition leg (eval forward [4,3,1024,2048] + scan), steady-state per-step table
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/pf && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf -o t -- python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 2 --warmup 1 --ramp 0 --acq-steps 8 > gpurun_out/pf_bench.json 2>/dev/null
python profiles/steady.py gpurun_out/pf/t_kernel_trace.csv k_cosine_fwd 5 gpurun_out/pool_forward_steady.md "rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 2 --warmup 1 --ramp 0 --acq-steps 8" > /dev/null
head -36 gpurun_out/pool_forward_steady.md | cut -c1-130
