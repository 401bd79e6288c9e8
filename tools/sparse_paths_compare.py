"""The same sparse LP (bench.py's sparse-lp construction with l samples) through four paths -- the tiled copy under the one-pass and
the carried schedule, round 5's two CSR copies, the dense-ified matrix -- tau, the first criterion and the sizes of the iterate after
1, 2, 5, 10, 20, 45 iterations side by side (round 6: this is how "tau sits at its clamp from iteration 5 on" was seen to be the
problem's, not a path's).      python tools/sparse_paths_compare.py 4096"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import importlib.util
spec=importlib.util.spec_from_file_location('bench','bench.py'); b=importlib.util.module_from_spec(spec); sys.argv=['x']; spec.loader.exec_module(b)
import totsu_amd as T
from totsu_amd import _lib
_lib.init()
l=int(sys.argv[1]) if len(sys.argv)>1 else 4096
i=b.sparse_lp_instance(l)
n,m=i['n'],i['m']
A=sp.csc_matrix((i['vals'],i['rowidx'],i['colptr']),shape=(m,n))
p=T.SolverParam(); p.eps_acc=0.0; p.eps_inf=0.0; p.max_iter=None
res={}
for name,kw,sched in (("tile-sweep",{}, "sweep"),("tile-carried",{}, "carried"),("csr-carried",{"sparse_two_copies":True},"carried"),("dense-carried",None,"carried")):
    if kw is None:
        fs=T.FusedSolver(n,m,np.asfortranarray(A.toarray()).ravel(order="F"),i['b'],i['c'],i['seg_type'],i['seg_len'],p,sched)
    else:
        fs=T.FusedSolver(n,m,A,i['b'],i['c'],i['seg_type'],i['seg_len'],p,sched,**kw)
    for k in (1,2,5,10,20,45):
        pass
    out=[]
    done=0
    for k in (1,2,5,10,20,45):
        r=fs.run(k-done,poll_every=64); done=k
        x,y=fs.iterate()
        out.append((k, r.tau, r.cri[0], float(np.abs(x).max()), float(np.abs(y).max())))
    res[name]=out
    print(name, fs.schedule_in_use())
    for o in out: print("   ",o)
    fs.destroy()
