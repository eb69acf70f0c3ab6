set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s33; mkdir -p $O
timeout -k 10 200 python tools/s2k3_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/s2k3.log
for rep in 1 2; do for L in libmulactseg_hip.so libvar_wxstag.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_train_table.py --out $O/bx_train_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
done; done
