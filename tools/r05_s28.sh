set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s28; mkdir -p $O
V=$PWD/mulactseg_amd/libvar_sh16.so
MAS_LIB=$V timeout -k 10 300 python tools/sh16_debug.py > $O/debug.log 2>&1; cat $O/debug.log
