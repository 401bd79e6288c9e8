"""The f64 oracle's answer for bench.py --workload sparse-sdp (partitioning_sdp construction, PSD order k) through its sparse
user-operator, to eps_acc: the objective a GPU line is compared with (profiles/r06_sparse_sdp_oracle_objective.json).
    python tools/sparse_sdp_oracle_objective.py 500 1e-3"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(bench)
sys.argv = argv
k, eps = int(sys.argv[1]), float(sys.argv[2])
inst = bench.sparse_sdp_instance(k)
t0 = time.time()
r = O.solve_csc_cones(O.param(max_iter=400000, eps_acc=eps), inst["c"], inst["colptr"], inst["rowidx"], inst["vals"].astype(np.float64),
                      inst["b"], inst["seg_type"], inst["seg_len"], use_ql=True)
out = {"workload": inst["what"], "eps_acc": eps, "status": int(r.status), "iterations": int(r.iters) + 1,
       "primal_objective": float(inst["c"].astype(np.float64) @ r.x), "dual_objective": -float(inst["b"].astype(np.float64) @ r.y),
       "seconds": time.time() - t0, "threads": O.num_threads()}
print(json.dumps(out))
