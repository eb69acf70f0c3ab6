# rocprofv3 kernel stats of the stage-1 train step (bench.py's train_iter leg; the scan leg is kept minimal)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr -o t -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp 0 --train-steps 10 > gpurun_out/tr_bench.json 2>/dev/null
python profiles/summarize.py stats gpurun_out/tr/t_kernel_stats.csv gpurun_out/tr_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp 0 --train-steps 10"
head -50 gpurun_out/tr_stats.md
