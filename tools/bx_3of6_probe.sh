# What halving the MFMA count of k_conv_bx buys on a power-limited chip (a measurement build with WRONG results: -DBX_PROBE_3OF6):
#   bash tools/build_variant.sh conv_bx "-DBX_PROBE_3OF6" libbx_3of6.so; bash tools/bx_3of6_probe.sh
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for L in product libbx_3of6.so; do
  if [ "$L" = product ]; then unset MAS_LIB; else export MAS_LIB=$PWD/build/variants/$L; fi
  echo "== $L"
  python tools/layer_alone.py --cin 64 --cout 64 --h 512 --w 1024 2>&1 | grep "us per"
  python tools/layer_alone.py --cin 512 --cout 512 --h 64 --w 128 --dil 2 2>&1 | grep "us per"
  python tools/layer_alone.py --cin 2048 --cout 512 --h 64 --w 128 --k 1 2>&1 | grep "us per"
  python tools/layer_alone.py --cin 64 --cout 256 --h 256 --w 512 --k 1 2>&1 | grep "us per"
  python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 2 --warmup 1 --ramp 0 --acq-steps 24 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pool batch %.2f ms' % d['pool_forward_ms_per_batch'])"
done; done
