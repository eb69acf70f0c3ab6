set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s5; mkdir -p $O
timeout -k 10 600 python tools/bx_pre_table.py --shape pool --out $O/bx_pre_pool.md > $O/pre_pool.log 2>&1; echo "pre pool rc $?"; tail -3 $O/pre_pool.log
timeout -k 10 300 python tools/bx_pre_table.py --shape train --out $O/bx_pre_train.md > $O/pre_train.log 2>&1; echo "pre train rc $?"; tail -3 $O/pre_train.log
for rep in 1 2; do for L in libmulactseg_hip.so libvar_stag1.so libvar_stag2.so; do
  MAS_LIB=$PWD/mulactseg_amd/$L timeout -k 10 300 python tools/bx_table.py --out $O/bx_table_${L%.so}_$rep.md 2>/dev/null | tail -2 | sed "s/^/$L: /"
done; done > $O/stagger_ab.log 2>&1; cat $O/stagger_ab.log
