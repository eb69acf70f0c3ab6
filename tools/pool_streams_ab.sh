# A/B of MAS_POOL_STREAMS on the acquisition leg of bench.py in ONE GPU session (same box): bash tools/pool_streams_ab.sh [values...]
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for n in "${@:-1 2 3}"; do
    for v in $n; do
      MAS_POOL_STREAMS=$v python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 2 --warmup 1 --ramp 0 --acq-steps 24 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MAS_POOL_STREAMS=$v', 'pool batch %.2f ms' % d['pool_forward_ms_per_batch'], 'mfma frac %.3f' % d['pool_forward_mfma_frac'])"
    done
  done
done
