set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s40; mkdir -p $O
MAS_LIB=$PWD/mulactseg_amd/libvar_rese.so timeout -k 10 600 python -m pytest tests/test_conv_bx_gpu.py tests/test_aspp_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for rep in 1 2; do for L in libmulactseg_hip.so libvar_rese.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 400 python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 3 --warmup 1 --acq-steps 16 > $O/bench_${L%.so}_$rep.json 2> /dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/s40/bench_${L%.so}_$rep.json').read().strip().split('\n')[-1])
print('$L pool forward ms/batch', d.get('pool_forward_ms_per_batch'))
PY
  echo "$L train:"; MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop 768 2>&1 | grep -E "^own" | tail -1
done; done
