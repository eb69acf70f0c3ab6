# PMC passes over single geometries of the training convolution kernels: where do the cycles go
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/pmcsk
rm -rf $O; mkdir -p $O
i=0
for shape in "1024 256 1 1 1 4 48 48 20 fwd" "256 1024 1 1 1 4 48 48 20 fwd" "512 512 3 1 1 4 48 48 10 fwd" "256 256 1 1 1 4 192 192 10 fwd" "1024 256 1 1 1 4 48 48 20 wgrad"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a$i -o a -- python tools/sk_probe.py $shape > $O/a$i.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $O/b$i -o b -- python tools/sk_probe.py $shape > $O/b$i.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/c$i -o c -- python tools/sk_probe.py $shape > $O/c$i.log 2>&1 || true
done
python - <<'PY'
import csv, collections, glob
for d in sorted(glob.glob('gpurun_out/pmcsk/[abc]?')):
    f = glob.glob(d + '/*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'k_conv_sk' in r['Kernel_Name'] or ('k_wgrad<' in r['Kernel_Name']):
            acc['k'][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(d, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
cat $O/a?.log | grep "us,"
