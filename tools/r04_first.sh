#!/bin/bash
# round 4, first GPU call: the new tests, then the bench line
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_counts or config4 or bench_streams or register or walk_form" > "$OUT/r04a_parity.log" 2>&1; tail -5 "$OUT/r04a_parity.log"
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "mixed_rate" > "$OUT/r04a_fuzz.log" 2>&1; tail -5 "$OUT/r04a_fuzz.log"
timeout 900 python3 -m pytest tests/test_gpu_sonic2.py tests/test_gpu_speedy_unit.py tests/test_gpu_cli.py tests/test_gpu_pool.py -m gpu -x -q > "$OUT/r04a_api.log" 2>&1; tail -5 "$OUT/r04a_api.log"
timeout 900 python3 bench.py > "$OUT/r04a_bench.json" 2> "$OUT/r04a_bench.err"; tail -c 600 "$OUT/r04a_bench.err"; python3 -c "
import json
d=json.load(open('$OUT/r04a_bench.json'))
print('value',d['value'],'ms',d['ms_per_step'])
r=d['roofline']; print('hbm frac',r['frac'],'kernels',r['kernel_ms_per_step'])
print('latency',r['latency'])
print('valu',{k:v for k,v in (r['valu_fp64'] or {}).items() if k in ('standalone_ms','achieved','frac','flop_per_frame')})
for k in ('large_batch','pcie_inclusive','config4_shard','config4_full','api_256_handles'):
    v=d.get(k); print(k, {a:b for a,b in v.items() if a not in ('note','kernels')} if v else None)
c=d.get('cpu_baseline'); print('cpu',{a:b for a,b in c.items() if a not in ('sample',)} if c else None)
"
