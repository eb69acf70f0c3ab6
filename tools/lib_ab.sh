# A/B of library builds on the scan leg of bench.py in ONE GPU session (same box, same clocks):
#   bash tools/lib_ab.sh libA.so libB.so ...     (names under build/variants/, or 'product' for the in-tree library; each twice, interleaved)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = product ]; then unset MAS_LIB; else export MAS_LIB=$PWD/build/variants/$L; fi
    python bench.py --no-cpu-baseline --no-pool --no-train --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$L', 'value %.2f M/s' % (d['value'] / 1e6), 'ms/step %.4f' % d['ms_per_step'], 'kernel', {k: r[k] for k in r if k in ('achieved', 'frac', 'kernel_us', 'us_per_launch')})"
  done
done
