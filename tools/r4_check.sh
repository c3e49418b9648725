#!/bin/bash
# round 4: the new configs[4] tests and the bench line with its workloads block
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g | head -2; nproc
timeout 1500 python -m pytest tests/test_gpu_config_size.py -x -q -m gpu -s -k "50k_exemplars or 50000-400" 2>&1 | grep -v "^$" | tail -12
timeout 900 python bench.py > gpurun_out/bench_r4a.txt 2> gpurun_out/bench_r4a.err; tail -c 600 gpurun_out/bench_r4a.err; python3 - <<'PY'
import json
for l in open('gpurun_out/bench_r4a.txt'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'steps', 'ms_per_step_after_device_sync', 'uberblast_e2e_ms', 'get_similar_pairs_ms') if k in d})
        print(d.get('map_workload'))
        c = d.get('cpu_baseline') or {}
        print({k: c.get(k) for k in ('value', 'kind', 'cores', 'cpu_model', 'phase_s', 'sw_cells_per_s', 'sw_cells_per_s_vectorised', 'gpu_hits_identical')})
        w = d.get('workloads') or {}
        print(json.dumps(w, indent=None)[:3000])
PY
