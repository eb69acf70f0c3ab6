set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/p5
timeout -k 10 900 bash tools/profile_r05.sh train > gpurun_out/p5/train.log 2>&1; echo "train rc $?"; tail -30 gpurun_out/p5/train.log | cut -c1-160
timeout -k 10 300 bash tools/profile_r05.sh loss > gpurun_out/p5/loss.log 2>&1; echo "loss rc $?"; tail -12 gpurun_out/p5/loss.log | cut -c1-160
