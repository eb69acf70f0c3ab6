set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/pmcsq
for mode in 1 0; do
export MAS_SINGLE_PASS_RING=$mode
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a$mode -o a -- python tools/kbench.py fused > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_LDS --output-format csv -d $O/b$mode -o b -- python tools/kbench.py fused > /dev/null 2>&1
done
find $O -name "*.csv" | head -20
