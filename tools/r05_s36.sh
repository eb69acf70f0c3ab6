set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s36; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_aspp_gpu.py tests/test_train_golden.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
python - <<'PY'
import torch, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
from mulactseg_amd import ops, _lib
from conv_table import timeit
lib=_lib.load()
for (N,C,H,W,d) in ((4,2048,64,128,6),(4,2048,48,48,6),(4,2048,49,49,6)):
    x=torch.randn(N,C,H,W,device='cuda'); ws=[torch.randn(C,1,3,3,device='cuda') for _ in range(3)]
    ys=[torch.empty_like(x) for _ in range(3)]
    st=torch.cuda.current_stream().cuda_stream
    f=lambda: lib.mas_aspp_dw3_fwd(x.data_ptr(),ws[0].data_ptr(),ws[1].data_ptr(),ws[2].data_ptr(),N,C,H,W,d,2*d,3*d,ys[0].data_ptr(),ys[1].data_ptr(),ys[2].data_ptr(),st)
    t=timeit(f)
    ref=[torch.nn.functional.conv2d(x,ws[j],None,1,(j+1)*d,(j+1)*d,groups=C) for j in range(3)]
    err=max(float((ys[j]-ref[j]).abs().max()) for j in range(3))
    dx=torch.empty_like(x)
    g=lambda: lib.mas_aspp_dw3_bwd_x(ys[0].data_ptr(),ys[1].data_ptr(),ys[2].data_ptr(),ws[0].data_ptr(),ws[1].data_ptr(),ws[2].data_ptr(),N,C,H,W,d,2*d,3*d,dx.data_ptr(),st)
    t2=timeit(g)
    gb=x.numel()*4*4/1e9
    print((N,C,H,W,d),"fwd %.1f us (%.2f TB/s) err %.2e | bwd_x %.1f us"%(t,gb/t*1e3,err,t2))
PY
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 3 --warmup 1 --acq-steps 12 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/s36/bench.json').read().strip().split('\n')[-1])
print('pool forward ms/batch', d.get('pool_forward_ms_per_batch'))
PY
