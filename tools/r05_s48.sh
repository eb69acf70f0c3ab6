set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s48; mkdir -p $O
for rep in 1 2; do for L in libmulactseg_hip.so libvar_wxp2.so libvar_wxp2b.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_train_table.py --out $O/bx_train_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /" | cut -c1-40,150-260
done; done
