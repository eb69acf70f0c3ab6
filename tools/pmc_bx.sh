# PMC passes over single convolution geometries on csrc/conv_bx.hip: where do the cycles of the split-bf16 kernel go
#   bash tools/pmc_bx.sh            (inside a gpurun call; summaries under gpurun_out/pmcbx)
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/pmcbx
rm -rf $O; mkdir -p $O
i=0
for shape in "1024 256 1 1 1 4 64 128" "256 256 3 1 1 4 64 128" "64 256 1 1 1 4 256 512" "512 2048 1 1 1 4 64 128"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a$i -o a -- python tools/conv_probe.py $shape 10 bx > $O/a$i.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b$i -o b -- python tools/conv_probe.py $shape 10 bx > $O/b$i.log 2>&1
done
python - <<'PY'
import csv, collections, glob
for d in sorted(glob.glob('gpurun_out/pmcbx/[ab]?')):
    f = glob.glob(d + '/*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'k_conv_bx' in r['Kernel_Name']:
            acc['conv'][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(d, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
cat $O/a?.log | grep bx
