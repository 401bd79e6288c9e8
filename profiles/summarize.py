"""Turns rocprofv3 result databases (gpurun_out/*/…_results.db) into the text summaries committed here.

  python profiles/summarize.py stats gpurun_out/prof_r01/bench_results.db > profiles/r01_socp_carried_kernel_stats.txt
  python profiles/summarize.py pmc gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db KEY
  python profiles/summarize.py counters gpurun_out/pmc_a/x_results.db [more.db ...]
  python profiles/summarize.py mfma <MfmaUtil.db> <raw counters.db> 500 profiles/r05_sdp_k500_mfma_util.json

`pmc` prints per-kernel averages of FETCH_SIZE / WRITE_SIZE (KiB as rocprofv3 reports them) and updates
profiles/hbm_traffic.json[KEY] with the corrected HBM bytes per launch of the dominant kernel: on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) => read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken
as reported (uncalibrated, and < 0.2 % of the traffic here).
"""
import json
import os
import sqlite3
import sys


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats  (durations in microseconds)   source: %s" % db)
    print("%-8s %-14s %-12s %-8s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel"))
    for name, calls, tot, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        print("%-8d %-14.1f %-12.2f %-8.3f  %s" % (calls, tot, avg, pct, short))


def counters(*dbs):
    """per (counter, kernel): launches, average counter value, average kernel duration"""
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = ("select counter_name, kernel_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by counter_name, kernel_name order by counter_name, sum(duration) desc")
        print("# source: %s" % db)
        for cname, name, calls, val, dur in cur.execute(q):
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if calls < 20:
                continue
            print("%-30s %-44s launches %-6d avg %-12.5g avg_us %.2f" % (cname, short, calls, val, dur / 1e3))


def mfma(util_db, raw_db, order, out_json):
    """profiles/r05_sdp_k500_mfma_util.json (bench.py's roofline_eig.MfmaUtil_stored): MfmaUtil per chain kernel from one --pmc
    pass, and from a second pass the raw counters behind it (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) with the busy fraction
    over the kernel's OWN duration (busy cycles / (SIMDs x duration x 2.4 GHz)): the derived counter divides by the profiler's
    window around the dispatch, about twice a 10 us kernel (NOTEBOOK.md 9.2)"""
    out = {"order": int(order), "source": "rocprofv3 --kernel-trace --pmc MfmaUtil / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                                          "-- python bench.py --workload sdp --no-cpu --no-to-eps --steps 6 --warmup 2",
           "files": [util_db, raw_db], "kernels": {}}
    cur = sqlite3.connect(util_db).cursor()
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = 'MfmaUtil' "
         "group by kernel_name order by sum(duration) desc")
    for name, calls, val, dur in cur.execute(q):
        if calls < 20 or not ("gemm_pre" in name or "polar_" in name):
            continue
        out["kernels"][_short(name)] = {"launches": calls, "MfmaUtil_percent": val, "avg_us_under_pmc": dur / 1e3}
    if raw_db and os.path.exists(raw_db):
        cur = sqlite3.connect(raw_db).cursor()
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by kernel_name, counter_name")
        for name, cname, calls, val, dur in cur.execute(q):
            k = _short(name)
            if k in out["kernels"]:
                out["kernels"][k][cname] = val
                out["kernels"][k]["avg_us_raw_pass"] = dur / 1e3
        for k, r in out["kernels"].items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in r:
                r["busy_fraction_over_kernel_duration"] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * r["avg_us_raw_pass"] * 1e-6 * 2.4e9)
                if r.get("GRBM_GUI_ACTIVE"):
                    # (the database holds the SUM over the 8 XCDs' instances; the derived counter takes their max)
                    r["window_us_at_2.4GHz"] = r["GRBM_GUI_ACTIVE"] / 8.0 / 2.4e3
                    r["MfmaUtil_recomputed_percent"] = 100.0 * r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


def _short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc(fetch_db, write_db, key):
    """FETCH_SIZE and WRITE_SIZE of ONE kernel instance: the matching instance (sweep_k<...> / dual_gemv_k<.., DO_N, DO_T, ..>)
    with the most TIME in the FETCH pass (launches x duration) -- the loop's kernel, not a geometry the plan autotune timed nor
    the 200 short sweeps of the publish-scope self-test -- and the SAME
    instance's row of the WRITE pass (round 3 paired the first row of each pass: two different instances when the two runs'
    autotunes disagreed)."""
    rows = {}
    for label, db in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
             "where counter_name = ? group by kernel_name order by sum(duration) desc")
        print("# rocprofv3 --pmc %s   source: %s" % (label, db))
        print("%-6s %-16s %-12s %s" % ("calls", "avg_KiB", "avg_us", "kernel"))
        rows[label] = {}
        for name, calls, val, dur in cur.execute(q, (label,)):
            short = _short(name)
            print("%-6d %-16.1f %-12.1f %s" % (calls, val, dur / 1e3, short))
            rows[label][short] = (calls, val, dur / 1e3)
        print()

    if key.startswith("sparse"):
        # the tiled sparse products: two instances per iteration (sp_tile_k<true> = A^T [v x_y], <false> = A [u x_x']); bench.py's
        # roofline averages both launches, and so does the stored traffic
        per, tot_rd, tot_wr, k_ = {}, 0.0, 0.0, 0
        # (the full-size instances: <T, LITE = false>; a LITE instance appears once per solve, for the |A| sums)
        for inst in ("thip::sp_tile_k<true, false>", "thip::sp_tile_k<false, false>"):
            f_, w_ = rows["FETCH_SIZE"].get(inst), rows["WRITE_SIZE"].get(inst)
            if f_ is None:
                continue
            rd, wr = 2.0 * f_[1] * 1024.0, (w_[1] if w_ else 0.0) * 1024.0
            per[inst] = {"launches_fetch_pass": f_[0], "fetch_size_KiB_raw": f_[1], "write_size_KiB_raw": w_[1] if w_ else None,
                         "read_bytes_corrected_x2": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "avg_us_fetch_pass": f_[2]}
            tot_rd += rd; tot_wr += wr; k_ += 1
        if not k_:
            print("# %s: no sp_tile_k in the FETCH pass" % key)
            return
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hbm_traffic.json")
        d = json.load(open(path)) if os.path.exists(path) else {}
        d[key] = {"kernel": "sp_tile_k<true> and sp_tile_k<false>, mean of the two launches of an iteration", "instances": per,
                  "hbm_bytes_per_launch": (tot_rd + tot_wr) / k_}
        json.dump(d, open(path, "w"), indent=1)
        print("# %s: corrected HBM bytes per launch (mean of both products) = %.4g" % (key, (tot_rd + tot_wr) / k_))
        return

    def wanted(short):
        if key.endswith("_sweep"):
            return "sweep_k<" in short
        return short.startswith("dual_gemv_k<") and ", true, true, false" in short and ("unsigned short" in short) == key.endswith("_bf16")
    cand = [(v[0] * v[2], k) for k, v in rows["FETCH_SIZE"].items() if wanted(k)]
    if not cand:
        print("# %s: no matching kernel in the FETCH pass" % key)
        return
    inst = max(cand)[1]
    fetch = rows["FETCH_SIZE"][inst]
    write = rows["WRITE_SIZE"].get(inst)
    if write is None:
        print("# %s: instance %s is not in the WRITE pass (another geometry was autotuned there): WRITE_SIZE left out" % (key, inst))
    rd = 2.0 * fetch[1] * 1024.0
    wr = (write[1] if write else 0.0) * 1024.0
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hbm_traffic.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = {"kernel": inst, "launches_fetch_pass": fetch[0], "launches_write_pass": write[0] if write else 0,
              "fetch_size_KiB_raw": fetch[1], "write_size_KiB_raw": write[1] if write else None,
              "read_bytes_corrected_x2": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
              "avg_us_fetch_pass": fetch[2]}
    json.dump(d, open(path, "w"), indent=1)
    print("# %s: %s: corrected HBM bytes per launch = 2*FETCH + WRITE = %.4g" % (key, inst, rd + wr))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "counters":
        counters(*sys.argv[2:])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
