set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s17; mkdir -p $O
rm -rf $O/t_tr
rocprofv3 --kernel-trace --output-format csv -d $O/t_tr -o t -- python tools/train_step_probe.py --modes own --streams main --steps 8 --crop 769 > $O/train_probe_769.log 2>&1
python profiles/steady.py $O/t_tr/t_kernel_trace.csv multi_tensor_apply 6 $O/d_train_769_steady.md "rocprofv3 --kernel-trace -- python tools/train_step_probe.py --modes own --streams main --steps 8 --crop 769" > /dev/null
head -45 $O/d_train_769_steady.md
timeout -k 10 600 python tools/bx_train_table.py --shape train769 --out $O/bx_train_table_769.md > $O/table769.log 2>&1; echo "table rc $?"; cat $O/bx_train_table_769.md
