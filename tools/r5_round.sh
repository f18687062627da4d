#!/bin/bash
# Round-5 GPU collection (gpurun -- bash tools/r5_round.sh TAG [parts]): the new tests, the bench line, the PCIe leg ten times,
# batch sizes with and without the sub-batch split, the C pipeline example with 4 and 8 hardware queues, counters and references.
TAG=${1:-r5b}
PARTS=${2:-tests,bench,pcie,scale,example,refs,sq}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
has() { case ",$PARTS," in *",$1,"*) return 0;; *) return 1;; esac; }
if has tests; then
  timeout 2400 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ahead.py tests/test_gpu_cli.py tests/test_gpu_dist.py tests/test_gpu_perf_guard.py \
    "tests/test_gpu_parity.py::test_committed_kernel_resources_are_the_librarys" -m gpu -x -q > "$OUT/${TAG}_pytest_new.log" 2>&1
  tail -5 "$OUT/${TAG}_pytest_new.log"
fi
if has bench; then
  timeout 900 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; tail -c 600 "$OUT/${TAG}_bench.err"
  python3 tools/bench_summary.py "$OUT/${TAG}_bench.json" 2>/dev/null | head -40
fi
if has pcie; then
  : > "$OUT/${TAG}_pcie_repeat.txt"
  for i in 1 2 3 4 5 6 7 8 9 10; do
    timeout 300 python3 bench.py --pcie-child 0,0,200 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('run $i: %.3f ms per batch  intervals %s  h2d alone %.3f ms  mode %d form %d' % (d['dt']*1e3, {k: round(v,3) for k,v in d['interval_ms'].items()}, d['h2d_alone_s']*1e3, d['last_mode'], d['walk_form']))" >> "$OUT/${TAG}_pcie_repeat.txt"
  done
  cat "$OUT/${TAG}_pcie_repeat.txt"
fi
if has scale; then
  { echo "# split into overlapping sub-batches (shipped)"; python3 tools/scale_streams.py 256 512 640 768 1024 2048
    echo "# one call (tuning build, SPX_SPLIT_MAX=1)"; SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so SPX_SPLIT_MAX=1 python3 tools/scale_streams.py 256 512 640 768 1024 2048
  } > "$OUT/${TAG}_scale_streams.txt" 2>&1
  cat "$OUT/${TAG}_scale_streams.txt"
fi
if has example; then
  python3 -c "
import sys; sys.path.insert(0, '.')
from speedy_amd.synth import speech_like
speech_like(160000, 16000, seed=3).astype('<i2').tofile('/tmp/in10s.raw')"
  { echo "# GPU_MAX_HW_QUEUES set by the library's constructor (8)"; for i in 1 2 3; do ./speedy_amd/lib/pipeline_example /tmp/in10s.raw 16000 1 3.5 1.0 256 300 4 /tmp/o.raw; done
    echo "# SPX_KEEP_HW_QUEUES=1 (HIP's default of 4)"; for i in 1 2 3; do SPX_KEEP_HW_QUEUES=1 ./speedy_amd/lib/pipeline_example /tmp/in10s.raw 16000 1 3.5 1.0 256 300 4 /tmp/o.raw; done
    echo "# depth 3"; ./speedy_amd/lib/pipeline_example /tmp/in10s.raw 16000 1 3.5 1.0 256 300 3 /tmp/o.raw
    echo "# depth 2"; ./speedy_amd/lib/pipeline_example /tmp/in10s.raw 16000 1 3.5 1.0 256 300 2 /tmp/o.raw
  } > "$OUT/${TAG}_pipeline_example.txt" 2>&1
  cat "$OUT/${TAG}_pipeline_example.txt"
fi
if has refs; then
  timeout 600 python3 tools/perf_reference.py | tail -1
  timeout 300 python3 tools/kernel_resources.py "$OUT/kernel_resources.json"
fi
if has sq; then
  bash tools/sq_counters.sh $TAG 2>&1 | tail -6
fi
ls "$OUT" | grep "$TAG" | head -40
