#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
bash tools/ab_variants.sh 2 probe1 probe2 probe3 probe4 probe5 probe6 probe7 probe8 probe9 probe10 > "$OUT/r04k_slack_probe.txt" 2>&1
cat "$OUT/r04k_slack_probe.txt"
