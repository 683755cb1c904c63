#!/bin/bash
# ms/step and search stats of a few workloads (run on the GPU box):  tools/wl.sh [env assignments]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
run() { env "$@" python3 "$ROOT/bench.py" --no-cpu 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_pass_ms']; s=d['search']
print('   ms/step %.3f search %.3f grid %.3f sums %.3f | cand %.0f retries %.0f fallback %s | %s' % (d['ms_per_step'], p['ms_search'], p['ms_grid'], p['ms_prep']+p['ms_density']+p['ms_pi']+p['ms_visc'], s['candidates_per_particle_step'], s['retries_per_step'], s.get('fallback_queries_last_step'), d['state_check']))"; }
for spec in "--workload polytrope" "--workload polytrope --forms loop" "--workload sedov --dt cfl" "--workload uniform_cube --forms loop" "--workload uniform_cube --steps 12" "--workload uniform_sphere --steps 10"; do
  echo "== $spec  [$*]"
  BENCH_ARGS="$spec" 
  env "$@" python3 "$ROOT/bench.py" --no-cpu $spec 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_pass_ms']; s=d['search']
print('   ms/step %.3f search %.3f grid %.3f sums %.3f | cand %.0f retries %.0f fallback %s | %s' % (d['ms_per_step'], p['ms_search'], p['ms_grid'], p['ms_prep']+p['ms_density']+p['ms_pi']+p['ms_visc'], s['candidates_per_particle_step'], s['retries_per_step'], s.get('fallback_queries_last_step'), d['state_check']))"
done
