#!/bin/bash
TAG=${1:-r04m}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
bash tools/gpu_round.sh $TAG quick 2>&1 | tail -8
bash tools/sq_counters.sh $TAG 2>&1 | tail -4
