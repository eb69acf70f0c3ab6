set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests/test_scorer_gpu.py -x -q -k lowres 2>&1 | tail -2
rm -rf gpurun_out/st && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st -o s -- python tools/scan_forms_probe.py > /dev/null 2>&1
grep "k_single_pass" gpurun_out/st/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
