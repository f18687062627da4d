#!/bin/bash
# GPU box: large-batch throughput of the walk-kernel variants.  bash tools/scale_sweep.sh "VARIANT1;VARIANT2" "512 1024 2048"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
IFS=';' read -ra VS <<< "${1:-;SPX_WALK_OLD=1}"
for v in "${VS[@]}"; do
  echo "== [$v]"
  env $v timeout 600 python3 tools/scale_streams.py ${2:-512 1024 2048} 2>&1 | grep streams
done | tee gpurun_out/scale_sweep.txt
