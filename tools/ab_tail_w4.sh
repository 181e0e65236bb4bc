#!/bin/bash
# A/B of the 4-wave tail-conv forms (S2SR_TAIL_W4=1) against the 8-wave forms: interleaved bench runs + per-family stats.
set -e
mkdir -p gpurun_out
for f in 1 0; do
  echo "--- per-family stats, S2SR_TAIL_W4=$f"
  S2SR_TAIL_W4=$f python tools/quick_bench.py --batch 512 --hp 1 --steps 3 | grep -E "SR-MP|conv_last|conv_hr|conv_up|conv_body"
done
for i in 1 2; do
  for f in 1 0; do
    S2SR_TAIL_W4=$f python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/ab_tw4_${f}_$i.json 2> gpurun_out/ab_tw4_${f}_$i.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_tw4_${f}_$i.json").read().strip().splitlines()[-1])
print("tail_w4=$f run $i:", d["value"], d["unit"], d["ms_per_step"], "ms/step")
PY
  done
done
