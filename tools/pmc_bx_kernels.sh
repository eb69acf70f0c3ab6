# What bounds the split-bf16 kernels: SQ counters per kernel (separate --pmc passes, kernel trace only) -> gpurun_out/pmcbx/*.csv and a table
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/pmcbx; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- python tools/bx_micro.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -o b -- python tools/bx_micro.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/c -o c -- python tools/bx_micro.py > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, sys
sys.path.insert(0, 'profiles')
from summarize import short
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmcbx/*/*counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if not (k.startswith('k_conv_bx') or k.startswith('k_wgrad_bx')):
            continue
        k = k + " grid " + r.get('Grid_Size', '?')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (k, r['Dispatch_Id']) not in seen and f.find('/a/') >= 0:
            seen.add((k, r['Dispatch_Id']))
            cnt[k] += 1
names = ['SQ_BUSY_CU_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_INSTS_LDS',
         'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_VMEM', 'SQ_INSTS_VALU_MFMA_MOPS_BF16',
         'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_WAVES', 'SQ_INSTS_SALU']
out = ["# SQ counters of the split-bf16 kernels (tools/pmc_bx_kernels.sh; sums over the launches of tools/bx_micro.py, three --pmc passes)", "",
       "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES).  Per-wave shares are over SQ_WAVE_CYCLES.", "",
       "| kernel | launches | MfmaUtil | VALU active / wave cyc | LDS active / wave cyc | wait LDS / wave cyc | wait any inst / wave cyc | wait any / wave cyc | LDS conflict / LDS idx active | VALU insts per MFMA-op | coexec / MFMA busy |",
       "|---|---|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(acc.items()):
    wc = max(v['SQ_WAVE_CYCLES'], 1.0)
    out.append("| %s | %d | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.2f | %.3f |" % (
        k[:80], cnt[k], v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(4 * v['SQ_BUSY_CU_CYCLES'], 1), v['SQ_ACTIVE_INST_VALU'] / wc, v['SQ_ACTIVE_INST_LDS'] / wc,
        v['SQ_WAIT_INST_LDS'] / wc, v['SQ_WAIT_INST_ANY'] / wc, v['SQ_WAIT_ANY'] / wc, v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1),
        v['SQ_INSTS_VALU'] / max(v['SQ_INSTS_VALU_MFMA_MOPS_BF16'], 1), v['SQ_VALU_MFMA_COEXEC_CYCLES'] / max(v['SQ_VALU_MFMA_BUSY_CYCLES'], 1)))
out += ["", "raw sums:", ""]
for k, v in sorted(acc.items()):
    out.append("* %s: %s" % (k[:80], ", ".join("%s %.4g" % (n, v[n]) for n in names if n in v)))
open('gpurun_out/pmcbx/q_bx_kernels_sq_pmc.md', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
