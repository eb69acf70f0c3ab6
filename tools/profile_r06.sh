# round 6 profiles -> gpurun_out/p6/*.md|json (copied into profiles/r06/ by hand).  Usage: bash tools/profile_r06.sh <part>
#   part scan   : bench.py scan leg: kernel stats + FETCH/WRITE PMC (roofline.traffic source)
#   part pool   : acquisition leg ONLY (eval forward [4,3,1024,2048] + quarter-resolution scan): steady table, MFMA PMC, HBM PMC
#   part train  : train step at 768 and 769 (own convolution kernels): steady tables, MFMA PMC, HBM PMC
#   part loss   : the stage-1 loss legs alone (tools/loss_probe.py): kernel stats + FETCH/WRITE PMC of k_partial_loss_fwd/bwd, full-resolution and quarter-resolution forms
set -eu; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
PART=${1:-scan}; O=gpurun_out/p6; mkdir -p $O
if [ "$PART" = scan ]; then
  CMD="python bench.py --no-cpu-baseline --no-train --no-pool"
  rm -rf $O/s_stats $O/s_fetch $O/s_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_stats -o s -- $CMD > $O/scan_bench.json 2> /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/s_fetch -o s -- $CMD --steps 20 --warmup 2 --ramp 0 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/s_write -o s -- $CMD --steps 20 --warmup 2 --ramp 0 > /dev/null 2>&1
  python profiles/summarize.py stats $(find $O/s_stats -name "*kernel_stats.csv") $O/a_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
  python profiles/summarize.py pmc $(find $O/s_fetch -name "*counter_collection.csv") $(find $O/s_write -name "*counter_collection.csv") $O/b_pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $CMD --steps 20 --warmup 2 --ramp 0"
  head -12 $O/a_bench_kernel_stats.md; head -c 900 $O/b_pmc_traffic.json
elif [ "$PART" = loss ]; then
  python tools/loss_probe.py --stepwise > $O/n_loss_probe.log 2>&1; python tools/loss_probe.py --crop 769 >> $O/n_loss_probe.log 2>&1; cat $O/n_loss_probe.log
  rm -rf $O/l_stats $O/l_fetch $O/l_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/l_stats -o l -- python tools/loss_probe.py --iters 30 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/l_fetch -o l -- python tools/loss_probe.py --iters 10 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/l_write -o l -- python tools/loss_probe.py --iters 10 > /dev/null 2>&1
  python profiles/summarize.py stats $(find $O/l_stats -name "*kernel_stats.csv") $O/n_loss_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/loss_probe.py --iters 30"
  python profiles/pmc_kernels.py $(find $O/l_fetch -name "*counter_collection.csv") $(find $O/l_write -name "*counter_collection.csv") $O/n_loss_hbm_pmc.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/loss_probe.py --iters 10"
  head -24 $O/n_loss_kernel_stats.md
elif [ "$PART" = pool ]; then
  CMD="python bench.py --no-cpu-baseline --no-pool --no-trainleg --steps 2 --warmup 1 --ramp 0 --acq-steps 8"
  rm -rf $O/p_tr $O/p_mf $O/p_fetch $O/p_write
  rocprofv3 --kernel-trace --output-format csv -d $O/p_tr -o t -- $CMD > $O/pool_bench.json 2> /dev/null
  python profiles/steady.py $O/p_tr/t_kernel_trace.csv k_cosine_fwd4 5 $O/e_pool_forward_steady.md "rocprofv3 --kernel-trace -- $CMD (acquisition leg only: no train / stage-2 leg in the process)" > /dev/null
  head -30 $O/e_pool_forward_steady.md
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -o f -- $CMD --acq-steps 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_write -o f -- $CMD --acq-steps 3 > /dev/null 2>&1
  python profiles/pmc_kernels.py $(find $O/p_fetch -name "*counter_collection.csv") $(find $O/p_write -name "*counter_collection.csv") $O/g_pool_forward_hbm_pmc.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $CMD --acq-steps 3"
  rm -rf $O/p_mf
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p_mf -o m -- $CMD --acq-steps 3 > /dev/null 2>&1
  python - "$CMD --acq-steps 3" <<'PY'
import csv, collections, sys, glob
sys.path.insert(0, 'profiles')
from summarize import short
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur, seen = collections.defaultdict(float), set()
for r in csv.DictReader(open(glob.glob('gpurun_out/p6/p_mf/*counter_collection.csv')[0])):
    k = short(r['Kernel_Name'])
    if k.startswith('k_conv') or k.startswith('k_stem'):
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
tot = collections.defaultdict(float)
rows = []
for k, v in acc.items():
    for c in v: tot[c] += v[c]
    rows.append((v['SQ_VALU_MFMA_BUSY_CYCLES'], k, v))
rows.sort(reverse=True)
out = ["# Matrix-pipe utilisation of the convolution kernels of the pool forward (rocprofv3 --pmc, all launches of the run)", "",
       "command: `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- %s`" % sys.argv[1], "",
       "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of the CU-busy cycles in which the matrix pipe of a SIMD is busy "
       "(a v_mfma_f32_32x32x16_bf16 holds it for 32 cycles, a v_mfma_f32_32x32x2_f32 for 64).", "",
       "all convolution kernels: MfmaUtil %.3f" % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 4 * tot['SQ_BUSY_CU_CYCLES'])), "",
       "Clock = GRBM_GUI_ACTIVE / 8 XCDs / the kernel's wall time (MI355X_MICROARCH.md, DVFS give-back: reads high on dispatches well under 0.3 ms): "
       "what the chip holds under this load, against the 2.4 GHz the 2.5 PFLOP/s bf16 figure is quoted at.", "",
       "| kernel | MfmaUtil | share of matrix-pipe cycles | clock (GHz) |", "|---|---|---|---|"]
for m, k, v in rows:
    out.append("| %s | %.3f | %.1f %% | %.2f |" % (k[:90], m / max(1.0, 4 * v['SQ_BUSY_CU_CYCLES']), 100 * m / max(1.0, tot['SQ_VALU_MFMA_BUSY_CYCLES']),
                                                  v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / max(1.0, dur[k])))
open('gpurun_out/p6/l_pool_forward_mfma_pmc.md', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
else
  for CROP in 768 769; do
    rm -rf $O/t_tr
    rocprofv3 --kernel-trace --output-format csv -d $O/t_tr -o t -- python tools/train_step_probe.py --modes own --streams async --steps 8 --crop $CROP > $O/train_probe_$CROP.log 2>&1
    python profiles/steady.py $O/t_tr/t_kernel_trace.csv k_adamw_multi 6 $O/c_train_${CROP}_steady.md "rocprofv3 --kernel-trace -- python tools/train_step_probe.py --modes own --streams async --steps 8 --crop $CROP" > /dev/null
    head -14 $O/c_train_${CROP}_steady.md; grep "^own" $O/train_probe_$CROP.log || true
  done
  rm -rf $O/t_mf $O/t_fetch $O/t_write
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/t_mf -o m -- python tools/train_step_probe.py --modes own --streams async --steps 4 > /dev/null 2>&1
  python - <<'PY'
import csv, collections, sys, glob
sys.path.insert(0, 'profiles')
from summarize import short
cc = list(csv.DictReader(open(glob.glob('gpurun_out/p6/t_mf/*counter_collection.csv')[0])))
disp = collections.OrderedDict()
for r in cc:
    e = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(disp)
marks = [i for i in ids if 'k_adamw_multi' in disp[i]['name']]
groups, last = [], None
for i in marks:
    if last is not None and i - last < 40 and groups: groups[-1] = i
    else: groups.append(i)
    last = i
a, b = groups[-4], groups[-1]
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
out = ["# MFMA utilisation of the stage-1 train step on this package's convolution kernels (3 steady steps, rocprofv3 --pmc)", "",
       "command: `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- python tools/train_step_probe.py --modes own --streams async --steps 4`", "",
       "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); MOPS_F32 x 512 = f32 MFMA FLOPs", "",
       "whole step: MfmaUtil %.3f, f32 MFMA FLOP %.3e per step" % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 4 * tot['SQ_BUSY_CU_CYCLES']), tot['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / 3), "",
       "| kernel | calls/step | MfmaUtil | share of MFMA cycles |", "|---|---|---|---|"]
for m, k, v in rows:
    out.append("| %s | %.1f | %.3f | %.1f %% |" % (k[:90], v['n'] / 3, m / max(1.0, 4 * v['SQ_BUSY_CU_CYCLES']), 100 * m / max(1.0, tot['SQ_VALU_MFMA_BUSY_CYCLES'])))
open('gpurun_out/p6/f_train_step_mfma_pmc.md', 'w').write("\n".join(out) + "\n")
print("\n".join(out[:26]))
PY
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/t_fetch -o f -- python tools/train_step_probe.py --modes own --streams async --steps 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/t_write -o f -- python tools/train_step_probe.py --modes own --streams async --steps 3 > /dev/null 2>&1
  python profiles/pmc_kernels.py $(find $O/t_fetch -name "*counter_collection.csv") $(find $O/t_write -name "*counter_collection.csv") $O/h_train_step_hbm_pmc.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/train_step_probe.py --modes own --streams async --steps 3"
fi
