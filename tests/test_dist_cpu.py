"""CPU, world_size 2, gloo: the row-sharded solve (totsu_amd/parallel.py) reproduces the single-process oracle --
same status, same iteration count, same iterates -- with one all-reduce of an n-vector per transposed product."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from problems import benchmark_lp, random_socp

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(tmp_path, case, port):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_worker.py"), str(tmp_path), case]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return [json.load(open(os.path.join(tmp_path, "rank%d.json" % r))) for r in range(2)]


def test_shard_segments_are_cone_aligned_and_cover():
    from totsu_amd.parallel import shard_segments
    from totsu_amd.synth import shard_cones
    seg = [100] * 10 + [1, 7, 0, 300]
    for world in (1, 2, 3, 4, 8):
        rows = []
        prev_s = 0
        for r in range(world):
            s0, s1, r0, r1 = shard_segments(seg, world, r)
            assert s0 == prev_s and r0 == sum(seg[:s0]) and r1 == sum(seg[:s1])
            prev_s = s1
            rows.append(r1 - r0)
        assert prev_s == len(seg) and sum(rows) == sum(seg)
        cs = [shard_cones(1000, world, r) for r in range(world)]
        assert cs[0][0] == 0 and cs[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(cs, cs[1:]))
        assert max(c1 - c0 for c0, c1 in cs) - min(c1 - c0 for c0, c1 in cs) <= 1


def test_sharded_socp_gloo(tmp_path):
    res = _launch(tmp_path, "socp", 29641)
    n, cones = 16, [4, 9, 0, 2, 7, 5]
    f, Gs, hs, cs, d = random_socp(n, cones, seed=11)
    ro = O.solve_socp(O.param(max_iter=20000, eps_acc=1e-7), f, Gs, hs, cs, d, np.zeros((0, n)), [], trace_cap=64)
    assert res[0]["status"] == res[1]["status"] == ro.status == O.OK
    assert res[0]["iters"] == res[1]["iters"] == ro.iters
    assert res[0]["x"] == res[1]["x"]                       # replicated vectors are identical on every rank
    assert np.allclose(res[0]["x"], ro.x, rtol=1e-8, atol=1e-10)
    y = np.concatenate([res[0]["y"], res[1]["y"]])
    assert res[0]["rows"][1] == res[1]["rows"][0] and res[1]["rows"][1] == sum(1 + k for k in cones)
    assert np.allclose(y, ro.y, rtol=1e-7, atol=1e-9)
    for a, b in zip(res[0]["trace_head"], ro.trace[:30]):
        assert a[0] == b[0] and np.allclose(a[2:], b[2:], rtol=1e-8, atol=1e-12)
    # 3 n-vector all-reduces per iteration (K^T y, K rx, criteria) + the sharded scalars
    per_iter = res[0]["collectives"] / (ro.iters + 1)
    assert 3 <= per_iter <= 8.5


def test_sharded_lp_gloo(tmp_path):
    res = _launch(tmp_path, "lp", 29642)
    c, G, h = benchmark_lp(20, seed=12)
    ro = O.solve_lp(O.param(max_iter=60000, eps_acc=1e-6), c, G, h, np.zeros((0, 20)), [], trace_cap=64)
    assert res[0]["status"] == res[1]["status"] == ro.status
    assert res[0]["iters"] == res[1]["iters"] == ro.iters
    assert np.allclose(res[0]["x"], ro.x, rtol=1e-7, atol=1e-9)
    assert np.allclose(np.concatenate([res[0]["y"], res[1]["y"]]), ro.y, rtol=1e-7, atol=1e-9)
