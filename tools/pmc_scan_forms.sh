# PMC passes over the two forms of the scan: VALU utilisation, issue stalls, LDS instructions (profiles/r02/k_scan_forms_pmc.md)
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/pmcscan
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- python tools/scan_forms_probe.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES --output-format csv -d $O/b -o b -- python tools/scan_forms_probe.py > $O/b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python tools/scan_forms_probe.py > $O/s.log 2>&1
python - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('a', 'b'):
    f = glob.glob('gpurun_out/pmcscan/%s/*counter_collection.csv' % d)[0]
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'lowres' if ('k_single_pass<' in n and 'true, true>' in n.replace(' ', '').replace(',', ', ')) else ('ring' if 'k_single_pass_ring' in n else None)
        if k is None and 'k_single_pass<' in n and n.rstrip().endswith('true>'):
            k = 'lowres'
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for r in csv.DictReader(open(glob.glob('gpurun_out/pmcscan/s/*kernel_stats.csv')[0])):
    if 'k_single_pass_ring' in r['Name']: dur['ring'] = float(r['AverageNs']) / 1e3
    elif 'k_single_pass<' in r['Name']: dur['lowres'] = float(r['AverageNs']) / 1e3
lines = ["# the two forms of the acquisition scan, batch [4,20,1024,2048] (rocprofv3 --pmc; tools/pmc_scan_forms.sh)", "",
         "ring = k_single_pass_ring on the materialised logits (671 MB); lowres = k_single_pass<LOWRES> on the quarter-resolution logits (42 MB), bilinear x4 in registers.",
         "Per wave: WAVE_CYCLES = ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY (quad-cycles); VALU share = ACTIVE_INST_VALU / WAVE_CYCLES.", "",
         "| form | avg us (un-profiled) | VALU instr / launch | LDS instr / launch | VALU share of wave cycles | issue stalls | parked (waitcnt / barrier) | LDS bank-conflict cycles |", "|---|---|---|---|---|---|---|---|"]
for k in ('ring', 'lowres'):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    wc = m.get('SQ_WAVE_CYCLES', 1.0)
    lines.append("| %s | %.1f | %.3g | %.3g | %.2f | %.2f | %.2f | %.3g |" % (k, dur.get(k, float('nan')), m.get('SQ_INSTS_VALU', 0), m.get('SQ_INSTS_LDS', 0),
                 m.get('SQ_ACTIVE_INST_VALU', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc, m.get('SQ_WAIT_ANY', 0) / wc, m.get('SQ_LDS_BANK_CONFLICT', 0)))
rows = 4 * 1024 * 8
for k in ('ring', 'lowres'):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    lines.append("")
    lines.append("%s: %.0f VALU + %.0f LDS wave-instructions per 256-pixel row (= %.1f + %.1f per pixel pair and class x 20 classes x 128 pairs / 64 lanes)" % (
        k, m.get('SQ_INSTS_VALU', 0) / rows, m.get('SQ_INSTS_LDS', 0) / rows, m.get('SQ_INSTS_VALU', 0) / rows / 40, m.get('SQ_INSTS_LDS', 0) / rows / 40))
open('gpurun_out/k_scan_forms_pmc.md', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
