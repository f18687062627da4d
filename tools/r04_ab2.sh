#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
TAG=$1; shift
bash tools/ab_variants.sh 4 "$@" > "$OUT/${TAG}_ab.log" 2>&1
cat "$OUT/${TAG}_ab.log"
