import numpy as np
from scipy.optimize import linprog
def best_quintic(l,u,deg=5):
    xs=np.unique(np.concatenate([np.geomspace(l,u,4000),np.linspace(l,u,4000)]))
    pw=[1,3,5][: (deg+1)//2]
    V=np.stack([xs**p for p in pw],1)
    k=len(pw)
    # minimize t: V c - 1 <= t ; 1 - V c <= t
    A=np.block([[V,-np.ones((len(xs),1))],[-V,-np.ones((len(xs),1))]])
    b=np.concatenate([np.ones(len(xs)),-np.ones(len(xs))])
    c=np.zeros(k+1); c[-1]=1
    r=linprog(c,A_ub=A,b_ub=b,bounds=[(None,None)]*k+[(0,None)],method="highs")
    co=r.x[:k]; E=r.x[-1]
    return co,E
l,u=1e-7,1.0
steps=[]
for it in range(20):
    co,E=best_quintic(l,u)
    steps.append((l,u,co,E))
    print(it,"l=%.3e u=%.6f"%(l,u),"coef",co,"E=%.3e"%E)
    l,u=1-E,1+E
    if E<3e-8: break
print("---- with safety margin")
l,u=1e-7,1.0
out=[]
for it in range(20):
    co,E=best_quintic(l,u*1.0005)
    out.append(co)
    # actual image of [l,u]
    xs=np.unique(np.concatenate([np.geomspace(l,u,20000),np.linspace(l,u,20000)]))
    p=co[0]*xs+co[1]*xs**3+co[2]*xs**5
    l,u=p.min(),p.max()
    print(it,"-> [%.9f, %.9f]"%(l,u))
    if max(1-l,u-1)<4e-7: break
for co in out: print("    { %.9ff, %.9ff, %.9ff }," % tuple(co))
import numpy as np
from scipy.optimize import linprog
def feasible(lo,hi,s):
    x1=np.linspace(0,hi,4001)[1:]
    V=lambda x: np.stack([x,x**3,x**5],1)
    A=[V(x1)]; b=[np.full(len(x1),hi)]                     # p <= hi
    x2=np.linspace(lo/s,hi,4001); A.append(-V(x2)); b.append(np.full(len(x2),-lo))   # p >= lo
    x3=np.linspace(0,lo/s,2001)[1:]; A.append(-V(x3)); b.append(-s*x3)             # p >= s x
    A=np.concatenate(A); b=np.concatenate(b)
    r=linprog(np.zeros(3),A_ub=A,b_ub=b,bounds=[(None,None)]*3,method="highs")
    return (r.status==0), (r.x if r.status==0 else None)
def best(lo,hi):
    a,bb=1.5,20.0; co=None
    for _ in range(40):
        m=(a+bb)/2
        ok,c=feasible(lo,hi,m)
        if ok: a=m; co=c
        else: bb=m
    return a,co
for lo,hi in [(0.7,1.2),(0.6,1.3),(0.5,1.5),(0.4,1.6),(0.3,1.7),(0.25,1.75),(0.2,1.8),(0.1,1.9)]:
    s,co=best(lo,hi)
    print("band [%.2f,%.2f] gain s=%.4f coef"%(lo,hi,s),co)
print("---- tail from [0.3,1.7]")
def best_quintic(l,u):
    xs=np.unique(np.concatenate([np.geomspace(l,u,3000),np.linspace(l,u,3000)]))
    V=np.stack([xs,xs**3,xs**5],1)
    A=np.block([[V,-np.ones((len(xs),1))],[-V,-np.ones((len(xs),1))]])
    b=np.concatenate([np.ones(len(xs)),-np.ones(len(xs))])
    c=np.zeros(4); c[-1]=1
    r=linprog(c,A_ub=A,b_ub=b,bounds=[(None,None)]*3+[(0,None)],method="highs")
    return r.x[:3],r.x[-1]
l,u=0.3,1.7
for it in range(5):
    co,E=best_quintic(l*0.98,u*1.01)
    xs=np.linspace(l,u,20001); p=co[0]*xs+co[1]*xs**3+co[2]*xs**5
    l,u=p.min(),p.max()
    print("    { %.9ff, %.9ff, %.9ff },   // -> [%.8f, %.8f]" % (co[0],co[1],co[2],l,u))
    if max(1-l,u-1)<1e-5: break
# lifting count
s=4.0616
import math
print("lifting steps from 1e-7*1.7:", math.log(0.3/(1.7e-7))/math.log(s))
