#!/bin/bash
# A/B of conv_last's folded 6-stage form against the 8-stage form: interleaved bench runs + per-family kernel stats.
set -e
mkdir -p gpurun_out
for i in 1 2; do
  for f in 1 0; do
    S2SR_LAST_FOLD=$f python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/ab_fold_${f}_$i.json 2> gpurun_out/ab_fold_${f}_$i.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_fold_${f}_$i.json").read().strip().splitlines()[-1])
print("fold=$f run $i:", d["value"], d["unit"], d["ms_per_step"], "ms/step")
PY
  done
done
for f in 1 0; do
  echo "--- per-family stats, S2SR_LAST_FOLD=$f"
  S2SR_LAST_FOLD=$f python tools/quick_bench.py --batch 512 --hp 1 --steps 3 | grep -E "SR-MP|conv_last|conv_hr|conv_up"
done
