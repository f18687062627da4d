#!/bin/bash
# GPU box: the at-scale comparisons with the oracle on the round's last build (the twiddle tables moved: every fp64 last bit of every
# transform is a new one) -- every spectrogram row and tension frame of 10 million frames per rate, the audio of 20 480 streams per
# rate, through plain calls, the pipeline object and mixed calls (tools/r11_probe.py).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6zp}
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
  for r in 8000 16000 22050 32000 44100 48000; do SPX_PROBE_RATE=$r timeout 900 python3 tools/r11_probe.py oracle 400 2>&1 | tail -1; done
  SPX_PROBE_RATE=16000 timeout 600 python3 tools/r11_probe.py oracle 40 2 1 0.1 1.5 2>&1 | tail -1
  SPX_PROBE_RATE=11025 timeout 600 python3 tools/r11_probe.py oracle 40 2>&1 | tail -1
  for r in 16000 22050 11025 44100; do SPX_PROBE_RATE=$r timeout 900 python3 tools/r11_probe.py audio 80 2>&1 | tail -1; done
  for r in 16000 22050; do SPX_PROBE_RATE=$r timeout 900 python3 tools/r11_probe.py pipeline 20 2>&1 | tail -1; done
  timeout 900 python3 tools/r11_probe.py mixed 40 2>&1 | tail -1
  timeout 900 python3 tools/r11_probe.py mixedpipe 40 2>&1 | tail -1
} | tee $OUT/${TAG}_final_build_probes.txt
