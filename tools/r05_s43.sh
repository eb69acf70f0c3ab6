set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s43; mkdir -p $O
timeout -k 10 300 python tools/aten_ops_probe.py > $O/aten.log 2>&1; tail -60 $O/aten.log | cut -c1-230
