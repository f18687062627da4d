#!/bin/bash
# On the GPU box: the library's HOST code under ASan + UBSan (make -C speedy_amd/csrc asan-host, built here or in the
# container; the device code is compiled as always) driving the real GPU through plain C programs over include/sonic2.h --
# tools/api_fuzz.c (random call sequences over 24-48 handles, both execution paths, which must agree byte for byte) and
# tools/stream_bench.c.  (Python + torch do not survive the sanitizer's HSA interceptors, so no pytest here.)
#   gpurun -- bash tools/asan_host.sh [seeds]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
[ -f speedy_amd/lib/asan/mixed_pipeline_example ] || make -s -C speedy_amd/csrc asan-host || exit 1
# (quarantine_size_mb: a program that frees tens of MB right before it exits makes the quarantine recycle chunks inside the ROCm
# runtime's own exit handlers, where the sanitizer's device allocator has already gone -- an internal CHECK of the sanitizer, not a report)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:quarantine_size_mb=4096 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
mkdir -p gpurun_out
LOG=gpurun_out/asan_host.log
: > $LOG
for seed in $(seq 1 ${1:-6}); do
  timeout 600 speedy_amd/lib/asan/api_fuzz $seed $((16 + 8 * (seed % 5))) 3000 >> $LOG 2>&1 || echo "api_fuzz seed $seed FAILED (rc $?)" >> $LOG
done
timeout 300 speedy_amd/lib/asan/stream_bench 64 4 >> $LOG 2>&1 || echo "stream_bench FAILED" >> $LOG
# (round 6) sixteen host threads in the reference's call order on the coalesced pool: flat combining, the lock released while the GPU works
timeout 300 speedy_amd/lib/asan/stream_bench 64 4 1000 3.5 1 threads:16 >> $LOG 2>&1 || echo "stream_bench threads:16 FAILED" >> $LOG
timeout 300 speedy_amd/lib/asan/stream_bench 48 2 700 1.5 1 threads:48 22050 >> $LOG 2>&1 || echo "stream_bench threads:48 FAILED" >> $LOG
# ... and the two C99 programs over include/speedy_hip.h: the batch call and the owning pipeline object (round 5), each checking its
# own outputs
RAW=gpurun_out/asan_in.raw
tail -c +45 tests/golden/tapestry.wav | head -c 96000 > $RAW     # 3 s of 16 kHz mono PCM16 behind the 44-byte header
if [ -f speedy_amd/lib/asan/batch_example ]; then
  timeout 600 speedy_amd/lib/asan/batch_example $RAW 16000 1 3.5 1 40 1 gpurun_out/asan_out1.raw >> $LOG 2>&1 || echo "batch_example FAILED (rc $?)" >> $LOG
fi
if [ -f speedy_amd/lib/asan/pipeline_example ]; then
  timeout 600 speedy_amd/lib/asan/pipeline_example $RAW 16000 1 3.5 1 64 9 4 gpurun_out/asan_out2.raw >> $LOG 2>&1 || echo "pipeline_example FAILED (rc $?)" >> $LOG
  timeout 600 speedy_amd/lib/asan/pipeline_example $RAW 16000 1 1.5 0 300 5 3 gpurun_out/asan_out3.raw >> $LOG 2>&1 || echo "pipeline_example (300 streams: sub-batches) FAILED (rc $?)" >> $LOG
  cmp gpurun_out/asan_out1.raw gpurun_out/asan_out2.raw >> $LOG 2>&1 || echo "batch and pipeline outputs differ FAILED" >> $LOG
fi
# (round 6) mixed-rate batches through the pipeline object, outputs on the device: the groups' walk kernels of consecutive batches overlap
if [ -f speedy_amd/lib/asan/mixed_pipeline_example ]; then
  timeout 600 speedy_amd/lib/asan/mixed_pipeline_example $RAW 96 11 4 >> $LOG 2>&1 || echo "mixed_pipeline_example FAILED (rc $?)" >> $LOG
  timeout 600 speedy_amd/lib/asan/mixed_pipeline_example $RAW 300 4 2 >> $LOG 2>&1 || echo "mixed_pipeline_example (300 streams) FAILED (rc $?)" >> $LOG
fi
cut -c1-220 $LOG | tail -12
echo "sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error\|FAILED' $LOG)"
