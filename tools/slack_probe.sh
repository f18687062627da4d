#!/bin/bash
# Build container: the probe variants of the library (about 100 cycles of s_nop at ONE site of the walk step each).
#   bash tools/slack_probe.sh "1 2 3 4 5 6 7 8 9 10"        then on the GPU box: bash tools/ab_variants.sh 2 probe1 probe2 ...
cd "$(dirname "$0")/.."
for k in $1; do
  bash tools/build_variant.sh probe$k "-DSPX_PROBE_SITE=$k" > /tmp/probe$k.log 2>&1 &
  while [ $(jobs -r | wc -l) -ge 3 ]; do sleep 2; done
done
wait
ls speedy_amd/lib/ab/
