#!/bin/bash
# The probes behind profiles/r05/ (gpurun -- bash tools/r5_probes.sh NAME [TAG]).  One function per file group; each prints what it
# measures and writes gpurun_out/TAG_*.txt.  The tuning build (make -C speedy_amd/csrc tuning) and the variant builds named below
# (tools/build_variant.sh NAME "-Dflag") must exist in the tree that is sent to the box.
NAME=${1:?usage: r5_probes.sh order|numa|queue|trace|host_sweep|log_ab|dft_ab|split|walk3|stamps [TAG]}
TAG=${2:-r5x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
V() { echo "$PWD/speedy_amd/lib/ab/libspeedy_hip_$1.so"; }
case $NAME in
  order)   # stream creation order and hardware queues: plain / three hand-rotated batches / pipeline resident / pipeline host to host
    { for o in lib_first pipe_first torch_first; do python3 tools/order_probe.py $o 4 2>/dev/null; done
      python3 tools/order_probe.py lib_first 3 2>/dev/null
      for o in lib_first pipe_first; do SPX_PROBE_QUEUES=4 python3 tools/order_probe.py $o 4 2>/dev/null; done
    } | tee "$OUT/${TAG}_order_probe.txt" ;;
  numa)    # where pinned pages land, H2D / D2H rates per placement policy
    { lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)"; rocm-smi --showtoponuma 2>/dev/null | grep -i numa; python3 tools/numa_probe.py; } 2>&1 | tee "$OUT/${TAG}_numa_probe.txt" ;;
  queue)   # the pipeline's run stream shifted among the hardware queues (tuning build)
    { echo -n "turns3: "; python3 tools/loop_trace.py turns3 2>/dev/null; echo -n "pipeline: "; python3 tools/loop_trace.py pipe_dev 2>/dev/null
      for d in 0 1 2 3 4; do echo -n "dummy streams $d: "; SPEEDY_HIP_LIB=$T SPX_PIPE_DUMMY_STREAMS=$d python3 tools/loop_trace.py pipe_dev 2>/dev/null; done
      echo -n "null stream: "; SPEEDY_HIP_LIB=$T SPX_PIPE_NULL_STREAM=1 python3 tools/loop_trace.py pipe_dev 2>/dev/null
    } | tee "$OUT/${TAG}_queue_probe.txt" ;;
  trace)   # kernel (+ memory copy) timelines of the loops
    for k in turns3 pipe_dev; do
      rocprofv3 --kernel-trace -d "$OUT/${TAG}_trace_$k" -o t --output-format csv -- python3 tools/loop_trace.py $k > "$OUT/${TAG}_trace_$k.log" 2>&1
      python3 tools/trace_summary.py $(find "$OUT/${TAG}_trace_$k" -name "*kernel_trace.csv" | head -1) 30 | tee "$OUT/${TAG}_trace_${k}_summary.txt"
      python3 tools/trace_step.py $(find "$OUT/${TAG}_trace_$k" -name "*kernel_trace.csv" | head -1) 24 > "$OUT/${TAG}_trace_${k}_last.txt"
    done
    rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/${TAG}_trace_host" -o t --output-format csv -- python3 tools/loop_trace.py pipe_host 120 > "$OUT/${TAG}_trace_host.log" 2>&1
    python3 tools/copy_trace_summary.py "$OUT/${TAG}_trace_host" 70 > "$OUT/${TAG}_trace_host_timeline.txt"; head -2 "$OUT/${TAG}_trace_host_timeline.txt" ;;
  host_sweep)  # buffer sets and gather-kernel widths of the host-to-host pipeline
    { for d in 4 5 6 8; do echo -n "depth $d: "; SPX_PROBE_DEPTH=$d python3 tools/loop_trace.py pipe_host 200 2>/dev/null; done
      for w in 8 16 32 64 128 256; do echo -n "gather workgroups $w, depth 6: "; SPEEDY_HIP_LIB=$T SPX_PIPE_PACK_WGS=$w SPX_PROBE_DEPTH=6 python3 tools/loop_trace.py pipe_host 200 2>/dev/null; done
      echo -n "no gather kernel: "; SPEEDY_HIP_LIB=$T SPX_PIPE_NO_GATHER=1 python3 tools/loop_trace.py pipe_host 200 2>/dev/null
    } | tee "$OUT/${TAG}_host_sweep.txt" ;;
  log_ab|dft_ab)  # analysis kernel alone: the shipped specs against the v1 build of ONE of them (variants logv1: -DSPX_LOG_V1, dftv1: -DSPX_DFT_V1)
    VN=$([ $NAME = log_ab ] && echo logv1 || echo dftv1)
    { for r in 1 2 3; do
        echo -n "shipped: "; python3 tools/analysis_time.py 16000 22050 48000 2>/dev/null | tr '\n' ' '; echo
        echo -n "$VN:   "; SPEEDY_HIP_LIB=$(V $VN) python3 tools/analysis_time.py 16000 22050 48000 2>/dev/null | tr '\n' ' '; echo
      done; } | tee "$OUT/${TAG}_${NAME}.txt" ;;
  split)   # batches of more streams than CUs: plain calls split or not (tuning: SPX_SPLIT_PLAIN), overlapped calls split or not (SPX_SPLIT_MAX)
    { echo "# plain calls, one call"; python3 tools/scale_streams.py 256 512 768 1024 2048
      echo "# plain calls, split (tuning build, SPX_SPLIT_PLAIN=4)"; SPEEDY_HIP_LIB=$T SPX_SPLIT_PLAIN=4 python3 tools/scale_streams.py 512 768 1024
      for s in 384 512; do
        echo -n "# pipeline, $s streams per batch, split: "; SPX_PROBE_STREAMS=$s python3 tools/loop_trace.py pipe_dev 30 2>/dev/null
        echo -n "# pipeline, $s streams per batch, one call (SPX_SPLIT_MAX=1): "; SPEEDY_HIP_LIB=$T SPX_SPLIT_MAX=1 SPX_PROBE_STREAMS=$s python3 tools/loop_trace.py pipe_dev 30 2>/dev/null
      done; } 2>&1 | tee "$OUT/${TAG}_split.txt" ;;
  walk3)   # a third walk stream (tuning build)
    { for r in 1 2; do echo -n "two walk streams: "; python3 tools/loop_trace.py pipe_dev 100 2>/dev/null
        echo -n "three walk streams: "; SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS3=1 python3 tools/loop_trace.py pipe_dev 100 2>/dev/null; done; } | tee "$OUT/${TAG}_walk3.txt" ;;
  stamps)  # the analysis kernel by phase (variant astamps: -DSPX_STAMPS, copied to speedy_amd/lib/stamps/)
    python3 tools/analysis_stamps.py 16000 2>&1 | tee "$OUT/${TAG}_analysis_stamps.txt" ;;
  *) echo "unknown probe $NAME"; exit 1 ;;
esac
