set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s37; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_aspp_gpu.py tests/test_train_golden.py tests/test_trainer_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
python - <<'PY'
import torch, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
from mulactseg_amd import ops, _lib
from conv_table import timeit
lib=_lib.load()
st=torch.cuda.current_stream().cuda_stream
for (N,C,H,W) in ((4,304,192,192),(4,256,192,192),(4,304,193,193),(2,8,20,36)):
    x=torch.randn(N,C,H,W,device='cuda'); g=torch.randn(N,C,H,W,device='cuda')
    part=torch.empty(N,C,9,device='cuda'); dw=torch.empty(C,1,3,3,device='cuda')
    f=lambda: lib.mas_depthwise3x3_bwd_w(x.data_ptr(),g.data_ptr(),N,C,H,W,1,part.data_ptr(),dw.data_ptr(),st)
    t=timeit(f)
    w=torch.zeros(C,1,3,3,device='cuda',dtype=torch.float64,requires_grad=True)
    y=torch.nn.functional.conv2d(x.double(),w,None,1,1,1,groups=C); y.backward(g.double())
    err=float((dw.double()-w.grad).abs().max()/w.grad.abs().max())
    print("dw bwd_w",(N,C,H,W),"%.1f us (%.2f TB/s) rel err %.2e"%(t,2*x.numel()*4/1e9/t*1e3,err))
for (N,C,H,W,d) in ((4,2048,48,48,6),(4,2048,49,49,6),(2,16,24,32,12)):
    x=torch.randn(N,C,H,W,device='cuda'); gs=[torch.randn(N,C,H,W,device='cuda') for _ in range(3)]
    dws=[torch.empty(C,1,3,3,device='cuda') for _ in range(3)]
    f=lambda: lib.mas_aspp_dw3_bwd_w(x.data_ptr(),gs[0].data_ptr(),gs[1].data_ptr(),gs[2].data_ptr(),N,C,H,W,d,2*d,3*d,dws[0].data_ptr(),dws[1].data_ptr(),dws[2].data_ptr(),st)
    t=timeit(f)
    err=0
    for j in range(3):
        w=torch.zeros(C,1,3,3,device='cuda',dtype=torch.float64,requires_grad=True)
        y=torch.nn.functional.conv2d(x.double(),w,None,1,(j+1)*d,(j+1)*d,groups=C); y.backward(gs[j].double())
        err=max(err,float((dws[j].double()-w.grad).abs().max()/w.grad.abs().max()))
    print("dw3 bwd_w",(N,C,H,W,d),"%.1f us rel err %.2e"%(t,err))
PY
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 769 2>&1 | grep -E "^own" | tail -1
