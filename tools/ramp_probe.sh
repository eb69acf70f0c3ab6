for r in 100 400 1500; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --ramp $r --no-cpu-baseline --no-train --no-pool 2>/dev/null > gpurun_out/ramp_$r.json
  python -c "
import json
d=json.load(open('gpurun_out/ramp_$r.json')); print('ramp', $r, round(d['value']/1e6,2), d['roofline']['avg_launch_ms'], d['config']['round_tail_ms'])"
done
