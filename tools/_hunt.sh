#!/bin/bash
timeout 200 python tools/_detect.py > gpurun_out/detect.txt 2>&1
cat gpurun_out/detect.txt | grep DETECT
if grep -q "DETECT uniform" gpurun_out/detect.txt; then
  timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -30
  timeout 400 python bench.py --no-configs --no-cpu-baseline 2>/dev/null > gpurun_out/bench_uniform.json
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_uniform.json").read().strip().splitlines()[-1])
print("BENCH", d["roofline"]["frac"], d["roofline"]["frac_of_copy"], d["pair_alloc_s"],
      [(a["candidates"], a["good_enough"], a["interleaved_bytes"] >> 30, round(a["search_ms"])) for a in d["pair_attempts"]],
      d["roofline_plain"]["frac"], d["roofline_own_input"]["frac"])
PY
fi
