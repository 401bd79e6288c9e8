"""GPU: the compiled hosts over the C ABI that stand in for the Rust crate (no cargo here):
examples/hip_prob_demo (the Hip* alias builders of include/totsu_f32hip_prob.hpp dispatching to the fused loop, and the
same problems call by call through the reference's composite operators and literal cones) and the trait-level host that
`bench.py --path trait` times."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_prob_alias_demo():
    exe = os.path.join(ROOT, "examples", "hip_prob_demo")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe, "3000", "400"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert sum("OK" in ln for ln in lines if ln.startswith("kat")) == 6 and not any("MISMATCH" in ln for ln in lines)
    # the mirror cache: 1 cone => f, G, h, c uploaded once each although as_op() is taken twice per G_i (socp.rs:450,463)
    assert any(ln.startswith("kat socp") and "uploads 4" in ln for ln in lines), r.stdout
    d = json.loads(lines[-1])
    assert d["ratio"] >= 0.9, d                      # the alias route IS the fused loop


@pytest.mark.parametrize("args", [["--workload", "lp", "--size", "600"], ["--workload", "socp", "--size", "300", "--cones", "6"]])
@pytest.mark.parametrize("cones", ["reference", "device"])
def test_trait_level_host_runs_and_reports(args, cones):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--path", "trait", "--trait-cones", cones,
                        "--steps", "40", "--no-cpu"] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["path"] == "trait" and d["value"] > 0 and d["trait_cones"] == cones


def test_f64_certificate_tool_on_a_small_instance():
    """tools/c3_f64_certificate.py (the full-size objective gate of bench.py's `objective_gate`) on n = 500: the GPU's
    f32 answer, evaluated in f64 on the host with the regenerated matrix, is primal feasible, dual feasible to eps and
    has a duality gap far inside 1e-4"""
    env = dict(os.environ, C3_N="500", C3_CONES="10")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c3_f64_certificate.py"), "1e-4"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    pt = json.loads(r.stdout.strip().splitlines()[-1])["points"][0]
    assert pt["state"] == 0
    assert pt["gap_rel"] <= 1e-4 and pt["dual_residual_rel"] <= 1.2e-4
    assert pt["primal_cone_violation_rel_to_norm_b"] <= 1e-5 and pt["dual_cone_violation_max"] <= 1e-5
    assert abs(pt["dual_residual_rel"] - pt["gpu_criteria_f32"][1]) <= 0.05 * pt["gpu_criteria_f32"][1] + 1e-6
