"""Parses the reference's golden trace (examples/nostd_cortex-m/log_qemu.txt, a DATA file produced by
examples/nostd_cortex-m/src/main.rs:57-99 under QEMU) into tests/golden/log_qemu.json.

Run in the dev container only (needs /root/reference):  python tests/golden/make_log_qemu_fixture.py
"""
import json
import os
import re

SRC = "/root/reference/examples/nostd_cortex-m/log_qemu.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "log_qemu.json")

out = {"source": "examples/nostd_cortex-m/log_qemu.txt", "problem": {
    # examples/nostd_cortex-m/src/main.rs:66-79 (column-major)
    "n": 2, "m": 3, "vec_c": [-1.0, 0.0], "mat_a_colmajor": [4.0, -1.0, -1.0, -1.0, 4.0, -1.0],
    "vec_b": [6.0, 6.0, 1.0], "cone": "rpos", "max_iter": 100000, "log_period": 10},
    "trace": []}
for line in open(SRC):
    m = re.match(r"query_worklen -> (\d+)", line)
    if m:
        out["query_worklen"] = int(m.group(1))
    m = re.match(r"\[DEBUG\] (\d+): pri_dual_gap (\S+) (\S+) (\S+)", line)
    if m:
        out["trace"].append({"iter": int(m.group(1)), "text": [m.group(2), m.group(3), m.group(4)]})
    m = re.match(r"solve -> \[(\S+), (\S+)\]", line)
    if m:
        out["x_text"] = [m.group(1), m.group(2)]
    if "Converged" in line:
        out["status"] = "Converged"
json.dump(out, open(DST, "w"), indent=1)
print("wrote", DST, len(out["trace"]), "records")
