# MFMA utilisation of the stage-1 train step (bench.py train_iter leg): PMC pass (kernel-trace + counters only)
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/mf && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mf -o m -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ramp 0 --train-steps 6 --acq-steps 1 > /dev/null 2>&1
python - <<'PY'
import csv, collections, sys
sys.path.insert(0, 'profiles')
from summarize import short
cc = list(csv.DictReader(open('gpurun_out/mf/m_counter_collection.csv')))
# dispatches in order; find optimizer markers to keep the last 3 train steps
disp = collections.OrderedDict()
for r in cc:
    d = int(r['Dispatch_Id'])
    e = disp.setdefault(d, {'name': r['Kernel_Name']})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(disp)
marks = [i for i in ids if 'multi_tensor_apply' in disp[i]['name']]
groups, last = [], None
for i in marks:
    if last is not None and i - last < 40 and groups: groups[-1] = i
    else: groups.append(i)
    last = i
a, b = groups[-4], groups[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for i in ids:
    if a < i <= b:
        e = disp[i]
        k = short(e['name'])
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'GRBM_GUI_ACTIVE'):
            acc[k][c] += e.get(c, 0.0)
        acc[k]['n'] += 1
tot = collections.defaultdict(float)
rows = []
for k, v in acc.items():
    for c in v: tot[c] += v[c]
    rows.append((v['SQ_VALU_MFMA_BUSY_CYCLES'], k, v))
rows.sort(reverse=True)
out = ["# MFMA utilisation of the stage-1 train step (3 steady steps, rocprofv3 --pmc)", "",
       "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); MOPS_F32 x 512 = f32 MFMA FLOPs", "",
       "whole step: MfmaUtil %.3f, f32 MFMA FLOP %.3e per step" % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 4 * tot['SQ_BUSY_CU_CYCLES']), tot['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / 3), "",
       "| kernel | calls/step | MfmaUtil | share of MFMA cycles |", "|---|---|---|---|"]
for m, k, v in rows[:16]:
    out.append("| %s | %.1f | %.3f | %.1f %% |" % (k[:90], v['n'] / 3, m / max(1.0, 4 * v['SQ_BUSY_CU_CYCLES']), 100 * m / max(1.0, tot['SQ_VALU_MFMA_BUSY_CYCLES'])))
open('gpurun_out/j_train_step_mfma_pmc.md', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
