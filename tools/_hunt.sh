#!/bin/bash
SMFFT_PAIR_NO_RESCAN=1 timeout 200 python tools/uniform_box_probe.py > gpurun_out/probe0.txt 2>&1
if grep -q "UNIFORM" gpurun_out/probe0.txt; then
  echo "RESULT uniform"
  grep -E "default budget|UNIFORM" gpurun_out/probe0.txt | cut -c1-420 | sed 's/^/NO_RESCAN /'
  timeout 200 python tools/uniform_box_probe.py 2>&1 | grep -E "default budget|UNIFORM|ordinary" | cut -c1-420 | sed 's/^/RESCAN /'
  timeout 200 python tools/uniform_box_probe.py 2>&1 | grep -E "default budget|UNIFORM|ordinary" | cut -c1-420 | sed 's/^/RESCAN2 /'
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "malloc or pacing or allocator or retired or written" 2>&1 | grep -E "passed|failed"
else
  echo "RESULT ordinary"
fi
