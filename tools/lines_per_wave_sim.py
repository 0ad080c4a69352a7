import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle as O, synth
offsets, pls = O.grid_offsets(desired_resolution=2048)
S = np.float32(np.log2(pls))
bf = synth.s_grid_init()[2]
o, d = synth.s_rays(0)
nears, fars = O.near_far_from_aabb(o, d, np.array([-1,-1,-1,1,1,1],np.float32), 0.2)
xyzs = O.march_rays_train(o, d, 1.0, bf, 1, 128, nears, fars, synth.s_noises(4096))[0]
kind = sys.argv[1] if len(sys.argv)>1 else 'stencil'
if kind=='stencil':
    e=np.float32(1e-2)
    offs=np.array([[0,0,0],[e,0,0],[-e,0,0],[0,e,0],[0,-e,0],[0,0,e],[0,0,-e]],np.float32)
    if len(sys.argv)>2 and sys.argv[2]=='interleaved':
        pts=np.clip(xyzs[:,None,:]+offs[None,:,:],-1,1).reshape(-1,3)   # sample-major: 7 stencil points adjacent
    else:
        pts=np.clip(xyzs[None,:,:]+offs[:,None,:],-1,1).reshape(-1,3)
elif kind=='uniform':
    pts=np.random.default_rng(0).uniform(-1,1,(1<<19,3)).astype(np.float32)
else:
    pts=xyzs
x=((pts+1)/2).astype(np.float32)
B=x.shape[0]; B=(B//64)*64; x=x[:B]
print(kind, 'B',B)
P1=np.uint32(2654435761); P2=np.uint32(805459861)
tot=0
for l in range(16):
    res=np.uint32(np.ceil(np.exp2(np.float32(l)*S)*np.float32(16)))
    size=int(offsets[l+1]-offsets[l])
    pos=np.minimum(np.maximum(x*np.float32(res)-np.float32(0.5),0),np.float32(res-1))
    pg=np.floor(pos).astype(np.uint32); pn=np.minimum(pg+1,res-1)
    stride=int(res); m1=m2=0
    if stride<=size: m1=stride; stride*=int(res)
    if stride<=size: m2=stride; stride*=int(res)
    hashed= stride>size
    lines_total=0; second_total=0
    for k in range(4):
        y = pn[:,1] if k&1 else pg[:,1]; z = pn[:,2] if k>>1 else pg[:,2]
        if hashed:
            yz=(y*P1)^(z*P2); r0=(pg[:,0]^yz)%size; r1=(pn[:,0]^yz)%size
        else:
            yz=y*np.uint32(m1)+z*np.uint32(m2); r0=(pg[:,0]+yz)%size; r1=(pn[:,0]+yz)%size
        line0=((r0.astype(np.int64)&~3)*4)>>7
        lw=line0.reshape(-1,64)
        ls=np.sort(lw,axis=1); nd=1+(np.diff(ls,axis=1)!=0).sum(1)
        lines_total+=nd.sum()
        need=((r0^r1)>=4)
        line1=(r1.astype(np.int64)*4)>>7
        l1=np.where(need,line1,-1).reshape(-1,64)
        ls=np.sort(l1,axis=1); nd1=(np.diff(ls,axis=1)!=0).sum(1)+ (ls[:,0]>=0)
        second_total+=nd1.sum()
    waves=B//64
    print(f"level {l:2d} res {int(res):5d} {'hash' if hashed else 'dense'} lines/wave: first {lines_total/waves:7.1f} second {second_total/waves:6.1f}")
    tot+=(lines_total+second_total)/waves
print('total lines per 64 points (all levels):', tot, ' x2.4 cyc =', tot*2.4)
