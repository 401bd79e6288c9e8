"""How many LIFTING steps of the polar chain (11 by default, gain 3.94 per step, band [0.3, 1.7]) does each PSD
projection of a solve actually need?  A numpy restatement of the conic iteration on an SDP of the bench's construction
(one PSD cone of order k, n variables) runs `iters` iterations in f64; for the input M of every projection it takes the
smallest relative eigenvalue |lambda| / ||M||_F above 2e-8 (below that a wrong sign costs less than f32 round-off) and
the number L of lifting steps that brings it into the band: 0.3 / (1.7 * 3.94^L) <= |lambda| / ||M||_F.
    python tools/psd_lift_histogram.py 200 800 3000        ->  profiles/r03_psd_lifting_steps_needed.txt
CPU only (numpy); evidence for why the chain's step count is not made adaptive."""
import numpy as np, math, sys
rng=np.random.default_rng(0)
k,n=int(sys.argv[1]),int(sys.argv[2]); iters=int(sys.argv[3])
sk=k*(k+1)//2
iu_r=np.array([r for c in range(k) for r in range(c+1)]); iu_c=np.array([c for c in range(k) for r in range(c+1)])
w=np.where(iu_r==iu_c,1.0,math.sqrt(2))
def svec(X): return X[iu_r,iu_c]*w
def smat(v):
    X=np.zeros((k,k)); X[iu_r,iu_c]=v/w; X[iu_c,iu_r]=v/w; return X
A=(rng.standard_normal((sk,n))/math.sqrt(k)).astype(np.float64)
x0=rng.standard_normal(n)/math.sqrt(n)
b=A@x0+svec(np.eye(k))
R=rng.standard_normal((k,k)); Yd=np.eye(k)+0.1/math.sqrt(k)*(R+R.T)/2
c=-A.T@svec(Yd)
m=sk
Tx=1/np.maximum(np.abs(A).sum(0)+np.abs(c),1e-12)
ty=np.abs(A).sum(1)+np.abs(b)
Ty=np.full(m,1/max(ty.max(),1e-12)); Ts=np.ones(m)
tt=np.abs(c).sum()+np.abs(b).sum()
Ttau=1/tt; Su=Tx.copy(); Sv=np.full(m,1/(ty.max()+1)); Sk=1/tt
xx=np.zeros(n); xy=np.zeros(m); xs=np.zeros(m); tau=1.0
u=np.zeros(n); v=np.zeros(m); kap=0.0
need=[]
def proj(vv):
    M=smat(vv)
    wv,Z=np.linalg.eigh(M)
    fro=np.linalg.norm(M)
    if fro>0:
        rel=np.abs(wv)/fro
        rel=rel[rel>2e-8]
        mn=rel.min() if rel.size else 1.0
        # lifting steps needed: 0.3/(1.7*3.94^L) <= mn
        L=max(0,math.ceil(math.log(0.3/(1.7*mn))/math.log(3.94)))
        need.append(L)
    wv=np.maximum(wv,0)
    return svec((Z*wv)@Z.T)
for it in range(iters):
    ox,oy,os_,ot=xx.copy(),xy.copy(),xs.copy(),tau
    xx=xx+Tx*(A.T@v + c*kap)
    xy=xy+Ty*(b*kap - A@u)
    xs=xs+Ts*v
    tau=max(tau+Ttau*(-(c@u)-(b@v)),0)
    xy=proj(xy); xs=proj(xs)
    rxx,rxy,rxs,rt=ox-2*xx,oy-2*xy,os_-2*xs,ot-2*tau
    u=u+Su*(-(A.T@rxy)-c*rt)
    v=v+Sv*(A@rxx+rxs-b*rt)
    kap=min(kap+Sk*(c@rxx+b@rxy),0)
need=np.array(need)
p=xs/tau-b+A@xx/tau; d=c+A.T@xy/tau
print("k",k,"cri",np.linalg.norm(p)/(1+np.linalg.norm(b)),np.linalg.norm(d)/(1+np.linalg.norm(c)))
print("histogram of lifting steps needed (per projection):",np.bincount(need,minlength=13))
print("by phase (first 10%, last 10%):",np.bincount(need[:len(need)//10],minlength=13),np.bincount(need[-len(need)//10:],minlength=13))
