set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/p5
timeout -k 10 900 bash tools/profile_r05.sh train > gpurun_out/p5/train.log 2>&1; echo "train rc $?"; head -9 gpurun_out/p5/c_train_768_steady.md | tail -4
