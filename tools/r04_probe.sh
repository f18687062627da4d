#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
{
for a in "alone" "dummy_streams 1" "dummy_streams 2" "dummy_streams 3" "dummy_streams 5" "main_conc" "main_serial" "chunks_set"; do
  python3 tools/probe_c4_prefix.py $a 2>&1 | grep shard
done
} > "$OUT/r04c_probe.log" 2>&1
cat "$OUT/r04c_probe.log"
