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


def _launch_sweep(tmp_path, case, port):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_sweep_worker.py"), str(tmp_path), case]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return [json.load(open(os.path.join(tmp_path, "rank%d.json" % r))) for r in range(2)]


@pytest.mark.parametrize("case,port", [("socp", 29651), ("lp", 29652)])
def test_column_sharded_sweep_gloo(tmp_path, case, port):
    """the N > 1 form of the one-pass schedule (column blocks, m-vectors replicated, ONE all-reduce per iteration),
    restated in numpy f64 (tests/sweep_numpy.py) and run on two gloo ranks: iterates and criteria of the reference
    iteration (the oracle) to f64 round-off -- the skewed order changes no value"""
    res = _launch_sweep(tmp_path, case, port)
    if case == "socp":
        n, cones = 16, [4, 9, 0, 2, 7, 5]
        f, Gs, hs, cs, d = random_socp(n, cones, seed=11)
        A = np.vstack([np.vstack([-c.reshape(1, n), -G]) for G, c in zip(Gs, cs)]).astype(np.float64)
        b = np.concatenate([np.concatenate([[dd], h]) for dd, h in zip(d, hs)]).astype(np.float64)
        c = f.astype(np.float64)
        seg_type, seg_len = [O.CONE_SOC] * len(cones), [1 + k for k in cones]
    else:
        c32, G, h = benchmark_lp(20, seed=12)
        A, b, c = G.astype(np.float64), h.astype(np.float64), c32.astype(np.float64)
        seg_type, seg_len = [O.CONE_RPOS], [40]
    n, m = c.size, b.size
    its = [1, 2, 10, 60]
    ro = O.solve_matop_cones(O.param(max_iter=100, eps_acc=1e-30), c, A, b, seg_type, seg_len,
                             snap_iters=[k - 1 for k in its], trace_cap=70)
    N = n + 2 * m + 1
    assert res[0]["cols"][1] == res[1]["cols"][0]
    for q, k in enumerate(its):
        s0, s1 = res[0]["snaps"][str(k)], res[1]["snaps"][str(k)]
        assert s0["xm"] == s1["xm"] and s0["ym"] == s1["ym"]        # the replicated m-vectors: identical on both ranks
        x = np.concatenate([s0["xx"], s1["xx"], s0["xm"]])
        y = np.concatenate([s0["u"], s1["u"], s0["ym"]])
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        assert np.allclose(x, rx, rtol=1e-9, atol=1e-12), (k, np.abs(x - rx).max())
        assert np.allclose(y, ry, rtol=1e-9, atol=1e-12), (k, np.abs(y - ry).max())
    for k in range(60):
        assert np.allclose(res[0]["cri"][k], ro.trace[k][2:], rtol=1e-8, atol=1e-13), (k, res[0]["cri"][k], ro.trace[k])
    # one all-reduce per sweep (61 sweeps for 60 iterations) + the one of the preconditioner
    assert res[0]["collectives"] == 62


def test_ranks_agree_on_column_shards_gloo(tmp_path):
    """a rank whose thip_sweep_probe says no makes EVERY rank take row shards (bench.py: agree_on_column_shards), whichever
    rank it is -- nobody is left alone in a column-sharded all-reduce"""
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29661", os.path.join(HERE, "dist_agree_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    res = [json.load(open(os.path.join(tmp_path, "rank%d.json" % r))) for r in range(2)]
    assert res[0] == res[1] == {"all_yes": True, "rank1_no": False, "rank0_no": False}


def test_kernel_resource_guard_fails_a_build_that_spills(tmp_path):
    """tools/check_kernel_resources.py (run by the Makefile on hipcc's -Rpass-analysis remarks): passes the budget the
    one-pass kernel has today, fails when an enforced instance spills beyond 16 bytes per lane or drops below two waves"""
    tool = os.path.join(os.path.dirname(HERE), "tools", "check_kernel_resources.py")

    def remarks(vgprs, spill, scratch, occ, inst="Li7ELi1ELi2ELi1ELi3ELi0E"):
        pre = "thip_sweep.hip:168:1: remark: "
        return "\n".join([pre + "Function Name: _ZN4thip7sweep_kI%sEEvNS_9SweepArgsE [-Rpass-analysis=kernel-resource-usage]" % inst,
                          pre + "    VGPRs: %d [-Rpass-analysis=kernel-resource-usage]" % vgprs,
                          pre + "    ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]" % scratch,
                          pre + "    Occupancy [waves/SIMD]: %d [-Rpass-analysis=kernel-resource-usage]" % occ,
                          pre + "    SGPRs Spill: 6 [-Rpass-analysis=kernel-resource-usage]",
                          pre + "    VGPRs Spill: %d [-Rpass-analysis=kernel-resource-usage]" % spill,
                          pre + "    LDS Size [bytes/block]: 656 [-Rpass-analysis=kernel-resource-usage]"]) + "\n"
    cases = [(remarks(255, 2, 12, 2), 0), (remarks(256, 14, 60, 2), 1), (remarks(128, 0, 0, 1), 1),
             (remarks(200, 0, 0, 2, inst="Li4ELi1ELi2ELi1ELi3ELi1E") + remarks(256, 40, 160, 2), 1),     # one bad instance among good ones
             ("no kernels here\n", 2)]
    for k, (txt, want) in enumerate(cases):
        f = tmp_path / ("r%d.txt" % k)
        f.write_text(txt)
        rc = subprocess.run([sys.executable, tool, str(f)], capture_output=True).returncode
        assert rc == want, (k, rc, want)
    # and the remarks of the library as built here pass
    built = os.path.join(os.path.dirname(HERE), "totsu_amd", "csrc", "thip_sweep.remarks.txt")
    if os.path.exists(built):
        assert subprocess.run([sys.executable, tool, built], capture_output=True).returncode == 0


def _bench_dry(args, env_extra=None, timeout=600):
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r, lines


def test_bench_gpus_8_control_path_at_the_real_shapes_without_a_gpu():
    """`python bench.py --gpus 8` -- the driver's command on a node nobody has had yet -- up to its first kernel, with real
    processes and real collectives (gloo): the ranks are spawned by bench.py itself, agree on column shards, plan their shards
    at the REAL configs[2] shape (100 000 x 50 000 over 8 column blocks + the row-sharded extra leg), assert the plan against
    288 GB, bracket a timed region with barriers and a max over ranks, and rank 0 prints ONE line carrying the keys of a real
    one.  --dry-run replaces the HIP library (no GPU here), nothing else."""
    r, lines = _bench_dry(["--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["value"] is None
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "rccl_ranks", "row_sharded", "time_to_eps", "objective_gate"):
        assert key in d, key
    plans = d["config"]["hbm_plan"]
    assert [p["rank"] for p in plans] == list(range(8))
    assert all(p["partition"] == "columns" and p["rows"] == 100_000 and p["cols"] == 6250 for p in plans)
    assert all(p["A_bytes"] == 4 * 100_000 * 6250 and p["row_leg_rows"] == 12_500 for p in plans)
    # the row leg's shard has columns of 50 000 B: the library's padded copy (12 512 rows) is in the plan
    assert all(p["row_leg_bytes"] == 4 * 12_500 * 50_000 + 4 * 12_512 * 50_000 for p in plans)
    assert all(p["fits"] and p["total_bytes"] < 0.1 * p["hbm_bytes"] for p in plans)
    assert d["row_sharded"]["rows_per_gpu"] == 12_500
    # the line names the partitioning behind `value` and carries both rates at top level (a SCALE record of the column form
    # must not read as north_star's row form)
    assert d["value_partitioning"].startswith("column-sharded") and "value_column_sharded" in d and "value_row_sharded" in d
    assert "bench.py: --gpus 8 without a launcher" in r.stderr


def test_bench_gpus_8_configs4_budget_and_a_rank_that_cannot_sweep():
    """configs[4] (LP n = 200 000, m = 400 000) over 8 ranks: a 40 GB column shard + the row leg's 40 GB shard per rank, inside
    288 GB; and when ONE rank's probe says no (rank 5), every rank plans row shards and the carried schedule"""
    r, lines = _bench_dry(["--gpus", "8", "--workload", "lp", "--size", "200000", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(lines[0])
    plans = d["config"]["hbm_plan"]
    assert all(p["partition"] == "columns" and p["rows"] == 400_000 and p["cols"] == 25_000 for p in plans)
    assert all(p["A_bytes"] == 40_000_000_000 and p["row_leg_bytes"] == 40_000_000_000 for p in plans)      # 50 000 rows: no padded copy
    assert all(p["fits"] and 80e9 < p["total_bytes"] < 0.94 * 288e9 for p in plans)
    r, lines = _bench_dry(["--gpus", "8", "--workload", "lp", "--size", "200000", "--steps", "2", "--warmup", "1"],
                          env_extra={"THIP_DRY_PROBE_NO": "5"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(lines[0])
    plans = d["config"]["hbm_plan"]
    assert all(p["partition"] == "rows" and p["rows"] == 50_000 and p["cols"] == 200_000 and p["row_leg_bytes"] == 0 for p in plans)
    assert d["config"]["schedule"] == "carried" and d["row_sharded"] is None
    assert d["value_partitioning"].startswith("row-sharded")
    assert "row shards, carried schedule" in r.stderr


def test_bench_plan_that_does_not_fit_stops_every_rank_before_allocating():
    """a shard that cannot fit 288 GB (LP n = 400 000 over 2 ranks: 640 GB each): every rank leaves with the plan in the
    message -- no rank is left in a collective, nothing was allocated"""
    r, lines = _bench_dry(["--gpus", "2", "--workload", "lp", "--size", "400000", "--steps", "2"], timeout=300)
    assert r.returncode != 0 and not lines
    assert "does not fit" in r.stderr


def test_bench_watchdog_prints_the_partial_line_when_a_rank_never_comes_back():
    """N > 1: a leg that hangs in a collective must not cost the line.  Rank 1 never reaches the barrier (test hook); after
    --watchdog seconds rank 0's timer thread -- the main thread sits inside the barrier's C call -- prints the line as far as it
    got, marked partial, and the launcher takes every rank down: the run ends instead of waiting for the driver's kill"""
    import time
    t0 = time.time()
    r, lines = _bench_dry(["--gpus", "2", "--steps", "2", "--watchdog", "6"], env_extra={"THIP_DRY_HANG_RANK": "1"}, timeout=240)
    assert time.time() - t0 < 200
    assert len(lines) == 1, (lines, r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["dry_run"] is True and "watchdog" in d and d["n_gpus"] == 2
    assert "watchdog on rank 0" in r.stderr
