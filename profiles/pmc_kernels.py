#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of the same
command, for EVERY kernel of this package (k_*), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of a
wide coalesced read; both counters are KiB).

  python profiles/pmc_kernels.py <fetch counter_collection.csv> <write counter_collection.csv> <out.md> "<command>" [skip_first_n]
"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from summarize import short  # noqa: E402


def main():
    fetch, write, out, cmd = sys.argv[1:5]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in (fetch, write):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_"):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# HBM traffic per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)", "", "command: `%s`" % cmd, "",
             "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction for wide coalesced reads; MI355X_MICROARCH.md, HBM section).", "",
             "| kernel | launches | fetch MB / launch (corrected) | write MB / launch | total MB / launch |", "|---|---|---|---|---|"]
    rows = []
    for k, c in acc.items():
        nf, nw = max(1, len(c["FETCH_SIZE"])), max(1, len(c["WRITE_SIZE"]))
        f = 2 * sum(c["FETCH_SIZE"]) / nf * 1024 / 1e6
        w = sum(c["WRITE_SIZE"]) / nw * 1024 / 1e6
        rows.append((sum(c["FETCH_SIZE"]) * 2 + sum(c["WRITE_SIZE"]), k, len(c["FETCH_SIZE"]), f, w))
    for _, k, n, f, w in sorted(rows, reverse=True):          # (every kernel: no row cut-off)
        lines.append("| %s | %d | %.1f | %.1f | %.1f |" % (k[:100], n, f, w, f + w))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
