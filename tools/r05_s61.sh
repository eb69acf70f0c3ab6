set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s61; mkdir -p $O
MAS_COSINE_CF=8 timeout -k 10 300 python -m pytest tests/test_head_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -1 $O/pytest.log
python - <<'PY'
import torch, sys, os, subprocess
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
PY
for cf in 4 8 4 8; do
MAS_COSINE_CF=$cf python - <<'PY'
import torch, sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
from mulactseg_amd import ops
from conv_table import timeit
for shape in ((4,256,192,192),(4,256,256,512)):
    f=torch.randn(*shape,device='cuda'); p=torch.randn(20,256,1,1,device='cuda')
    with torch.no_grad():
        t=timeit(lambda: ops.cosine_head(f,p))
    print("CF", os.environ.get("MAS_COSINE_CF"), shape, "%.1f us"%t)
PY
done
