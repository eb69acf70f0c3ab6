# rocprofv3 kernel-trace of tools/kbench.py fused; prints median duration of k_single_pass*
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/pf && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf -o t -- python tools/kbench.py fused "$@" > /dev/null 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/pf/**/t_kernel_trace.csv', recursive=True):
    du=sorted((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'k_single_pass' in r['Kernel_Name'])
    print("k_single_pass n=%d min %.1f p25 %.1f med %.1f p75 %.1f max %.1f us"%(len(du),du[0],du[len(du)//4],du[len(du)//2],du[3*len(du)//4],du[-1]))
PY
