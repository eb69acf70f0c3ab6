set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s60; mkdir -p $O
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -k 10 800 python tools/soak_conv_bx.py 500 9000 2>&1 | grep -v amdgpu.ids | tail -6
