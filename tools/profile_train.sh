# rocprofv3 kernel trace of the stage-1 train step and of the inference forward + scan (bench.py's secondary legs);
# steady-state per-step breakdowns (MIOpen's find phase excluded) -> gpurun_out/*.md
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/tr && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o t -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp 0 --train-steps 10 --acq-steps 8 > gpurun_out/tr_bench.json 2>/dev/null
python profiles/steady.py gpurun_out/tr/t_kernel_trace.csv multi_tensor_apply 6 gpurun_out/h_train_step_steady.md "rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp 0 --train-steps 10 --acq-steps 8 (train_iter leg)" > /dev/null
python profiles/steady.py gpurun_out/tr/t_kernel_trace.csv k_single_pass 5 gpurun_out/i_acquisition_forward_steady.md "same run, acquisition_with_model leg (eval forward [4,3,1024,2048] + scan)" > /dev/null
head -12 gpurun_out/h_train_step_steady.md; head -8 gpurun_out/i_acquisition_forward_steady.md
