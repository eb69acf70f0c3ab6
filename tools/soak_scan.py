import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from mulactseg_amd import ops, synth
from oracle import exact
bad=0
for seed in range(60):
    rs=np.random.RandomState(5000+seed)
    B=int(rs.randint(1,3)); C=int(rs.choice([19,20,21])); H=int(rs.randint(64,200)); W=int(4*rs.randint(1,160)); S=int(rs.randint(2,600))
    z=(rs.standard_normal((B,C,H,W))*rs.choice([0.3,1.0,3.0])).astype(np.float32)
    spx=np.stack([synth.superpixel_map(seed*7+i,H,W,S) for i in range(B)]).astype(np.int64)
    d=rs.uniform(size=spx.shape); spx[d<0.02]=-1; spx[(d>=0.02)&(d<0.04)]=S+1
    invT=ops.inv_temperature(0.1)
    eps_,ecs,eh=exact.single_pass_accum(z,spx,S,np.float32(invT))
    zt,st=torch.from_numpy(z).cuda(),torch.from_numpy(spx).cuda()
    for rep in range(3):
        ps,cs,hh=ops.single_pass_accum(zt,st,S,invT)
        ok=np.array_equal(ps.cpu().numpy().view(np.uint64),eps_) and np.array_equal(cs.cpu().numpy().view(np.uint64),ecs) and np.array_equal(hh.cpu().numpy().view(np.uint32),eh)
        if not ok: bad+=1; print('MISMATCH',seed,rep,B,C,H,W,S)
print('soak done, mismatches:',bad)
