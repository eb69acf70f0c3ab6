set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/inf && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/inf -o t -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp 0 --train-steps 1 --acq-steps 8 > /dev/null 2>&1
python profiles/steady.py gpurun_out/inf/t_kernel_trace.csv k_single_pass 5 gpurun_out/infer_steady.md "bench.py acquisition_with_model leg" | head -40
