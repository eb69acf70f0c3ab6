#!/usr/bin/env python
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small tracked summaries under profiles/.

  python profiles/summarize.py stats <kernel_stats.csv> <out.md> "<command that was profiled>"
  python profiles/summarize.py pmc <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<command>"

PMC correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced streaming read, WRITE_SIZE is exact; both are in KiB.  FETCH_SIZE and WRITE_SIZE are collected
in separate passes (TCC counter slots)."""
import collections
import csv
import os
import json
import re
import sys

OURS = ("k_class_prob_sum", "k_bvsb_region_accum", "k_region_finalize", "k_partial_loss_fwd", "k_partial_loss_bwd",
        "k_group_finalize", "k_loss_values", "k_loss_scales", "k_target_bits", "k_region_keys", "k_walk_cost", "k_walk_find",
        "k_walk_emit", "k_logits_iou", "k_iou_counts", "k_minmax", "k_single_pass", "k_aspp", "k_cosine")


def short(name):
    for k in OURS:
        if k in name:
            m = re.search(r"%s[a-z_0-9]*(<[^>]*>)?" % k, name)
            return m.group(0) if m else k
    m = re.search(r"\b(k_[a-z0-9_]+(<[^>]*>)?)", name)          # every other kernel of this package, template arguments kept
    if m:
        return m.group(1)
    m = re.search(r"(miopen\w+|MIOpen\w+|igemm_\w+|Cijk_\w+|naive_conv\w+|rocprim::\w+(::\w+)*|at::native::\w+(::\w+)*|\w+_kernel\w*)", name)
    return (m.group(1) if m else name)[:80]


def stats(path, out, cmd):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\ncommand: `%s`\n\n" % cmd)
        f.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write("| %s | %s | %.1f | %.1f | %.1f | %.2f | %s |\n" % (
                short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))


def pmc(fetch, write, out, cmd):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in (fetch, write):
        for r in csv.DictReader(open(path)):
            for k in OURS:
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = {}
    for rel in ("mulactseg_amd/csrc/single_pass.hip", "mulactseg_amd/csrc/scorer.hip", "mulactseg_amd/csrc/common.h", "mulactseg_amd/csrc/detmath.h"):
        srcs[rel] = hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()[:16]
    res = {"command": cmd, "kernel_source_sha16": srcs, "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts half of a "
                                          "wide coalesced read; MI355X_MICROARCH.md HBM section)", "kernels": {}}
    for k, c in acc.items():
        fs = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))
        ws = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
        res["kernels"][k] = {"launches": len(c["FETCH_SIZE"]), "FETCH_SIZE_KiB_raw": fs, "WRITE_SIZE_KiB_raw": ws,
                             "hbm_bytes_per_launch": (2 * fs + ws) * 1024}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:5])
    else:
        pmc(*sys.argv[2:6])
