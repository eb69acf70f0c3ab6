#!/usr/bin/env python
"""Steady-state per-step kernel breakdown from a rocprofv3 kernel trace (MIOpen's find phase and warm-up excluded).

  python profiles/steady.py <kernel_trace.csv> <marker substring> <n_last_steps> [out.md] ["command"]

A "step" ends at the last launch of a group of marker kernels (e.g. `multi_tensor_apply` = the optimizer step,
`k_single_pass` = the scan after an inference forward); the last n steps are averaged."""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from summarize import short  # noqa: E402


def main():
    path, marker, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    cmd = sys.argv[5] if len(sys.argv) > 5 else ""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ends, last = [], None
    for i, r in enumerate(rows):
        if marker in r['Kernel_Name']:
            t = int(r['Start_Timestamp'])
            if last is not None and t - last < 3e6 and ends:
                ends[-1] = i
            else:
                ends.append(i)
            last = t
    a, b = ends[-n - 1], ends[-1]
    wall = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e6 / n
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a + 1:b + 1]:
        k = short(r['Kernel_Name'])
        acc[k][0] += 1
        acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in acc.values())
    # kernels of two streams overlap (round 6: the weight gradients run beside the input-gradient chain): the union of the busy
    # intervals is the time in which SOME kernel runs, the sum of the durations counts overlapped time once per kernel
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows[a + 1:b + 1])
    union, cur_s, cur_e = 0, None, None
    for s_, e_ in iv:
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        union += cur_e - cur_s
    lines = ["# steady-state kernel breakdown (last %d steps, marker `%s`)" % (n, marker), "",
             "command: `%s`" % cmd, "",
             "wall %.2f ms/step, some kernel running %.2f ms/step, sum of kernel durations %.2f ms/step (kernels of two streams overlap)"
             % (wall, union / n / 1e6, tot / n / 1e3), "",
             "| kernel | calls/step | us/step | % |", "|---|---|---|---|"]
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):          # (every kernel of the step: no row cut-off)
        lines.append("| %s | %.1f | %.1f | %.1f |" % (k[:100], v[0] / n, v[1] / n, 100 * v[1] / tot))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
