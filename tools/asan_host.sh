#!/bin/bash
# On the GPU box: the library's host code under ASan + UBSan (make -C speedy_amd/csrc asan-host, built here or in the
# container) driving the real GPU -- the streaming API tests (both execution paths, the interleaved life-cycle fuzz) and the
# many-handle C program.  gpurun -- bash tools/asan_host.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
[ -f speedy_amd/lib/asan/libspeedy_hip.so ] || make -s -C speedy_amd/csrc asan-host || exit 1
A=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/asan/libspeedy_hip.so
mkdir -p gpurun_out
LD_PRELOAD=$A timeout 1500 python3 -m pytest tests/test_gpu_pool.py tests/test_gpu_sonic2.py tests/test_gpu_fuzz.py -m gpu -x -q \
  -k "not big_batch and not throughput and not differential" > gpurun_out/asan_host_pytest.log 2>&1
tail -5 gpurun_out/asan_host_pytest.log
LD_PRELOAD=$A speedy_amd/lib/asan/stream_bench 64 4 > gpurun_out/asan_host_stream_bench.log 2>&1; tail -3 gpurun_out/asan_host_stream_bench.log | cut -c1-300
grep -c "ERROR: AddressSanitizer\|runtime error" gpurun_out/asan_host_pytest.log gpurun_out/asan_host_stream_bench.log
