# round 2: kernel traces of the model legs (pool forward on the MFMA convolution; train step at 768 and 769) + MFMA PMC of the pool forward
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/tr2 && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr2 -o t -- python bench.py --no-cpu-baseline --no-pool --steps 2 --warmup 1 --ramp 0 --train-steps 8 --acq-steps 8 > gpurun_out/tr2_bench.json 2>/dev/null
python - <<'PY'
import csv, collections, sys
sys.path.insert(0, 'profiles')
from summarize import short
rows = list(csv.DictReader(open('gpurun_out/tr2/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# optimizer-step groups: the first run of train steps is crop 768, the second crop 769
ends, last = [], None
for i, r in enumerate(rows):
    if 'multi_tensor_apply' in r['Kernel_Name']:
        t = int(r['Start_Timestamp'])
        if last is not None and t - last < 3e6 and ends: ends[-1] = i
        else: ends.append(i)
        last = t
# two legs of (2 warm + 1 + 8) = 11 steps each
def table(a, b, n, title, out):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a + 1:b + 1]:
        k = short(r['Kernel_Name']); acc[k][0] += 1; acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in acc.values())
    wall = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e6 / n
    lines = ["# " + title, "", "wall %.2f ms/step, GPU busy %.2f ms/step" % (wall, tot / n / 1e3), "", "| kernel | calls/step | us/step | % |", "|---|---|---|---|"]
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
        lines.append("| %s | %.1f | %.1f | %.1f |" % (k[:100], v[0] / n, v[1] / n, 100 * v[1] / tot))
    open(out, 'w').write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))
print(len(ends), "optimizer steps")
if len(ends) >= 22:
    table(ends[4], ends[10], 6, "train step, crop 768: steady-state kernel breakdown (last 6 steps)", 'gpurun_out/c_train_768_steady.md')
    table(ends[15], ends[21], 6, "train step, crop 769: steady-state kernel breakdown (last 6 steps)", 'gpurun_out/d_train_769_steady.md')
PY
python profiles/steady.py gpurun_out/tr2/t_kernel_trace.csv k_cosine_fwd4 5 gpurun_out/e_pool_forward_steady.md "rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline --no-pool --steps 2 --warmup 1 --ramp 0 --train-steps 8 --acq-steps 8 (acquisition_with_model leg)" | head -30
rm -rf gpurun_out/mf2 && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mf2 -o m -- python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 2 --warmup 1 --ramp 0 --acq-steps 4 > /dev/null 2>&1
python - <<'PY'
import csv, collections, sys
sys.path.insert(0, 'profiles')
from summarize import short
cc = list(csv.DictReader(open('gpurun_out/mf2/m_counter_collection.csv')))
disp = collections.OrderedDict()
for r in cc:
    e = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(disp)
marks = [i for i in ids if 'k_cosine_fwd4' in disp[i]['name']]
# one cosine-head launch per forward; the acquisition leg's forwards come last: keep 3 whole steps before the last head launch
a, b = marks[-5], marks[-2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for i in ids:
    if a < i <= b:
        e = disp[i]; k = short(e['name'])
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'GRBM_GUI_ACTIVE'):
            acc[k][c] += e.get(c, 0.0)
        acc[k]['n'] += 1
tot = collections.defaultdict(float)
rows = []
for k, v in acc.items():
    for c in v: tot[c] += v[c]
    rows.append((v['SQ_VALU_MFMA_BUSY_CYCLES'], k, v))
rows.sort(reverse=True)
out = ["# MFMA utilisation of the pool forward + scan (3 steady steps, rocprofv3 --pmc)", "",
       "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); MOPS_F32 x 512 = f32 MFMA FLOPs", "",
       "whole step: MfmaUtil %.3f, f32 MFMA FLOP %.3e per step" % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 4 * tot['SQ_BUSY_CU_CYCLES']), tot['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / 3), "",
       "| kernel | calls/step | MfmaUtil | share of MFMA cycles |", "|---|---|---|---|"]
for m, k, v in rows[:12]:
    out.append("| %s | %.1f | %.3f | %.1f %% |" % (k[:90], v['n'] / 3, m / max(1.0, 4 * v['SQ_BUSY_CU_CYCLES']), 100 * m / max(1.0, tot['SQ_VALU_MFMA_BUSY_CYCLES'])))
open('gpurun_out/f_pool_forward_mfma_pmc.md', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
