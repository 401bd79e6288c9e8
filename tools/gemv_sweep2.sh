#!/bin/bash
# finer sweep: bash tools/gemv_sweep2.sh "<bench extra args>" "<nj list>" "<blocks list>" [steps]
EXTRA=$1; NJS=$2; BLS=$3; STEPS=${4:-30}
for nj in $NJS; do for blocks in $BLS; do
  out=$(THIP_GEMV_NT=1 THIP_GEMV_NJ=$nj THIP_GEMV_BLOCKS=$blocks python bench.py --steps $STEPS --warmup 3 --no-cpu $EXTRA 2>/dev/null)
  python - "$nj" "$blocks" "$out" <<'PY'
import json, sys
nj, blocks, out = sys.argv[1:4]
try:
    d = json.loads(out)
    print("nj=%s blocks=%-5s  gemv %.1f GB/s (%.3f ms)  iter/s %.2f" % (nj, blocks, d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["value"]))
except Exception as e:
    print("nj=%s blocks=%s FAILED %r" % (nj, blocks, e))
PY
done; done
