#!/bin/bash
# sweeps the dual-GEMV tuning knobs through bench.py (run on the GPU box); prints achieved GB/s per config
WL=${1:-socp}
STEPS=${2:-10}
EXTRA=${3:-}
for nt in 0 1; do for nj in 1 2 4; do for blocks in 1024 2048 4096 8192; do
  out=$(THIP_GEMV_NT=$nt THIP_GEMV_NJ=$nj THIP_GEMV_BLOCKS=$blocks python bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu $EXTRA 2>/dev/null)
  python - "$nt" "$nj" "$blocks" "$out" <<'PY'
import json, sys
nt, nj, blocks, out = sys.argv[1:5]
try:
    d = json.loads(out)
    print("nt=%s nj=%s blocks=%-5s  gemv %.1f GB/s (%.3f ms)  iter/s %.2f" % (nt, nj, blocks, d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["value"]))
except Exception as e:
    print("nt=%s nj=%s blocks=%s FAILED %r" % (nt, nj, blocks, e))
PY
done; done; done
