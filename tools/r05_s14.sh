set -u; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/s14; mkdir -p $O
rm -rf $O/t_tr
rocprofv3 --kernel-trace --output-format csv -d $O/t_tr -o t -- python tools/train_step_probe.py --modes own --streams main --steps 8 --crop 768 > $O/train_probe_768.log 2>&1
python profiles/steady.py $O/t_tr/t_kernel_trace.csv multi_tensor_apply 6 $O/c_train_768_steady.md "rocprofv3 --kernel-trace -- python tools/train_step_probe.py --modes own --streams main --steps 8 --crop 768" > /dev/null
cat $O/c_train_768_steady.md | head -70; grep "^own" $O/train_probe_768.log
for c in 768 769; do timeout -k 10 200 python tools/train_step_probe.py --modes own --streams main --steps 10 --crop $c 2>&1 | grep -E "^own" | tail -1; done
