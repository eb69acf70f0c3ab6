set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/p5
timeout -k 10 500 bash tools/profile_r05.sh scan > gpurun_out/p5/scan.log 2>&1; echo "scan rc $?"; tail -14 gpurun_out/p5/scan.log | cut -c1-200
timeout -k 10 600 bash tools/profile_r05.sh pool > gpurun_out/p5/pool.log 2>&1; echo "pool rc $?"; tail -30 gpurun_out/p5/pool.log | cut -c1-160
timeout -k 10 300 bash tools/pmc_bx_kernels.sh > gpurun_out/p5/pmcbx.log 2>&1; echo "pmcbx rc $?"; head -16 gpurun_out/pmcbx/q_bx_kernels_sq_pmc.md | cut -c1-220
