set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s50; mkdir -p $O gpurun_out/p5
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
