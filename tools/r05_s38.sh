set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/p5 gpurun_out/s38
timeout -k 10 700 bash tools/profile_r05.sh pool > gpurun_out/p5/pool.log 2>&1; echo "pool rc $?"; head -24 gpurun_out/p5/e_pool_forward_steady.md; cat gpurun_out/p5/l_pool_forward_mfma_pmc.md
timeout -k 10 500 bash tools/profile_r05.sh scan > gpurun_out/p5/scan.log 2>&1; echo "scan rc $?"; head -8 gpurun_out/p5/a_bench_kernel_stats.md
