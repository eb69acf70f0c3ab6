# MFMA / LDS / wait counters of single layers (tools/layer_alone.py): bash tools/layer_pmc.sh
set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/lp; rm -rf $O; mkdir -p $O
i=0
for L in "--cin 64 --cout 64 --h 512 --w 1024" "--cin 64 --cout 128 --h 512 --w 1024" "--cin 512 --cout 512 --h 64 --w 128 --dil 2" "--cin 64 --cout 64 --h 256 --w 512" "--cin 256 --cout 256 --h 64 --w 128"; do
  i=$((i+1))
  python tools/layer_alone.py $L
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/a$i -o a -- python tools/layer_alone.py $L > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/b$i -o b -- python tools/layer_alone.py $L > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/c$i -o c -- python tools/layer_alone.py $L > /dev/null 2>&1
  python - "$O" "$i" "$L" <<'PY'
import csv, glob, sys, collections
O, i, L = sys.argv[1:4]
acc = collections.defaultdict(float); n = 0
for d in "abc":
    fs = glob.glob("%s/%s%s/*counter_collection.csv" % (O, d, i))
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if "k_conv_bx" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(L, {k: "%.3g" % v for k, v in sorted(acc.items())})
if acc.get("SQ_BUSY_CU_CYCLES"):
    print("   MfmaUtil %.3f" % (acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * acc["SQ_BUSY_CU_CYCLES"])))
if acc.get("SQ_WAVE_CYCLES"):
    print("   LDS wait / wave cycles %.3f, bank conflict / LDS active %.3f" % (acc.get("SQ_WAIT_INST_LDS", 0) / acc["SQ_WAVE_CYCLES"], acc.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, acc.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
done
