#!/bin/bash
# the hipGraphLaunch crash after destroy + recapture of a shape (tests/test_model_gpu.py followed by tests/test_full_size_gpu.py): mitigations
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03r; mkdir -p $O
cd $R
for mode in empty_cache keep; do
  INSV2V_GRAPH_PURGE=$mode timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest_$mode.txt 2>&1
  echo "== INSV2V_GRAPH_PURGE=$mode rc=$?"; grep -E "passed|failed|Segmentation" $O/pytest_$mode.txt | head -3
done
