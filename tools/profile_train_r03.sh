# round 3: kernel trace of the stage-1 train step (tools/train_step_probe.py, one mode) -> steady-state per-step table
# usage: bash tools/profile_train_r03.sh <mode> <out.md> [crop]
set -eu
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
MODE=${1:-auto}; OUT=${2:-gpurun_out/train_steady.md}; CROP=${3:-768}
rm -rf gpurun_out/tr3 && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr3 -o t -- python tools/train_step_probe.py --modes "$MODE" --streams main --steps 8 --crop "$CROP" > gpurun_out/tr3_probe.log 2>&1
tail -2 gpurun_out/tr3_probe.log | head -1
python profiles/steady.py gpurun_out/tr3/t_kernel_trace.csv multi_tensor_apply 6 "$OUT" "rocprofv3 --kernel-trace -- python tools/train_step_probe.py --modes $MODE --streams main --steps 8 --crop $CROP" > /dev/null
head -60 "$OUT"
