#!/usr/bin/env python
"""Full-pool parity, with a number on it (SURVEY section 7 hard part (iii)).

The 2 975 x 2 048 synthetic pool of bench.py's pool round (mulactseg_amd.synth_pool: LogitSource + SyntheticPool + SyntheticLabels,
100 000 clicks under fair counting) through ``RegionSelector.select_next_batch`` on the HIP backend, then the SAME pool -- every
picture's logits and map fetched from the device -- through the two CPU oracles:

* ``oracle/exact.c`` (the detmath arithmetic the kernels implement): scores, class weights and the consumed prefix must be EQUAL,
  bit for bit (``hip_equals_exact_c``; the script fails otherwise);
* ``oracle/port.py`` (the reference's own f32 operation order on torch CPU: active_selection/my_bvsb_predclsbal_pwr_banignore.py:
  35-91, active_selection/base.py:27-38, dataloader/region_active_dataset.py:31-73): reported -- the symmetric difference of the two
  selected sets, max |delta score|, the gap between neighbouring scores at the cut-off, the class-weight delta, and for every
  flipped region its distance from the cut-off score.

    python tools/full_pool_parity.py [--pictures 2975] [--out gpurun_out/full_pool_parity.json]
    python tools/full_pool_parity.py --phase 1 --state gpurun_out/full_pool_state.npz      # exact.c + the reference's batch means
    python tools/full_pool_parity.py --phase 2 --state build/full_pool_state.npz           # the reference's pass 2 + the report
(two phases: one call on the GPU box is limited to 20 minutes and the full pool needs ~19 of host time; phase 1 leaves the batch
means, the class weights and the exact.c verdict in a small file that the second call reads)

Host time on the GPU box: ~0.2 s per picture for the reference order at 20 threads + the C oracle on 16 threads (~10 min for the
full pool).  A progress line is printed every 100 pictures."""
import argparse
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pictures", type=int, default=2975)
    ap.add_argument("--threads", type=int, default=min(20, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_pool_parity.json"))
    ap.add_argument("--phase", type=int, default=0, choices=[0, 1, 2])
    ap.add_argument("--state", default=os.path.join(ROOT, "gpurun_out", "full_pool_state.npz"))
    a = ap.parse_args()
    import test_pool_scale_gpu as T            # the pool builders and the numpy restatement of the tuple sort + budget walk
    from oracle import exact, port
    C, H, W, S = T.C, T.H, T.W, T.S
    n_img, batch = a.pictures, 4
    budget = int(T.BUDGET * n_img / T.N_POOL)
    t_all = time.time()
    pool, labels, net = T._make(n_img, {})
    scores, consumed, sel, dt = T._round(pool, labels, net, tempfile.mkdtemp(), budget)
    sc = scores.cpu().numpy()
    w_hip = sel.cls_weight.cpu().numpy()
    ci, cid, csc = T._consumed_arrays(consumed, pool)
    rank = np.arange(n_img)
    cost = labels.multi_hot_cls.sum(axis=2)
    valid = np.ones((n_img, S), np.uint8)
    print("HIP round: %d pictures, %d regions, %d selected for %d clicks in %.2f s" % (n_img, n_img * S, len(ci), budget, dt), flush=True)

    invT = exact.inv_temperature(0.1)
    exact.lib()
    torch.set_num_threads(a.threads)
    n_batches = (n_img + batch - 1) // batch

    def host_batch(lo):
        hi = min(lo + batch, n_img)
        return np.stack([net.host(i) for i in range(lo, hi)]), np.stack([pool.host_map(i) for i in range(lo, hi)])

    if a.phase == 2:
        st = np.load(a.state, allow_pickle=True)
        assert int(st['pictures']) == n_img
        means = [torch.from_numpy(m) for m in st['means']]
        eq = json.loads(str(st['eq']))
        ei = np.zeros(int(st['selected_exact_c']))
        assert np.array_equal(st['w_hip'], w_hip), "the HIP round of phase 2 differs from phase 1's"
    else:
        # pass 1 of both oracles on the same host copies: exact.c accumulators (one picture per worker thread), the reference's batch means
        ps, cs, hh, means = [], [], [], []
        t0 = time.time()
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            for b in range(n_batches):
                z, m = host_batch(b * batch)
                futs = [ex.submit(exact.single_pass_accum, z[k][None], m[k][None], S, invT) for k in range(z.shape[0])]
                means.append(port.class_prior_batch(torch.from_numpy(z), 0.1))
                for f in futs:
                    p = f.result()
                    ps.append(p[0]), cs.append(p[1]), hh.append(p[2])
                if (b + 1) % 25 == 0 or b + 1 == n_batches:
                    print("pass 1: %d / %d pictures, %.0f s" % (min((b + 1) * batch, n_img), n_img, time.time() - t0), flush=True)
        ps, cs, hh = np.concatenate(ps), np.concatenate(cs), np.concatenate(hh)
        batch_of = (np.arange(n_img) // batch).astype(np.int32)
        _, w_exact = exact.class_weight(ps, H * W, batch_of, n_batches, 6.0)
        esc, _, _ = exact.region_finalize_weighted(cs, hh, exact.weights_to_fixed31(w_exact), C - 1)
        ei, eid, escs, _ = T.numpy_select(esc, valid, rank, cost, budget)
        eq = {"class_weights": bool(np.array_equal(w_exact, w_hip)), "scores": bool(np.array_equal(esc, sc)),
              "consumed_prefix": bool(len(ei) == len(ci) and np.array_equal(ei, ci) and np.array_equal(eid, cid) and np.array_equal(escs, csc))}
        eq["score_elements_that_differ"] = int((esc != sc).sum())
        print("HIP vs oracle/exact.c:", eq, flush=True)
        if a.phase == 1:
            os.makedirs(os.path.dirname(a.state), exist_ok=True)
            np.savez(a.state, pictures=n_img, means=np.stack([m.numpy() for m in means]), eq=json.dumps(eq), selected_exact_c=len(ei),
                     w_hip=w_hip, seconds=time.time() - t_all)
            print("phase 1 written to", a.state, flush=True)
            if not (eq["class_weights"] and eq["scores"] and eq["consumed_prefix"]):
                sys.exit("HIP differs from oracle/exact.c")
            return

    # pass 2 in the reference's f32 operation order
    _, wref = port.class_weight(means, 6.0)
    ref = np.empty((n_img, S), dtype=np.float32)
    t0 = time.time()
    for b in range(n_batches):
        z, m = host_batch(b * batch)
        r, h = port.region_scores_batch(torch.from_numpy(z), torch.from_numpy(m), 0.1, wref, S, C)
        r, _ = port.ban_ignore_dominant(r.view(-1), h.view(-1, C))
        ref[b * batch:b * batch + z.shape[0]] = r.view(-1, S).numpy()
        if (b + 1) % 25 == 0 or b + 1 == n_batches:
            print("pass 2: %d / %d pictures, %.0f s" % (min((b + 1) * batch, n_img), n_img, time.time() - t0), flush=True)
    ri, rid_, rsc, _ = T.numpy_select(ref, valid, rank, cost, budget)
    ours, theirs = set(zip(ci.tolist(), cid.tolist())), set(zip(ri.tolist(), rid_.tolist()))
    sym = sorted(ours ^ theirs)
    nz = ref != 0
    rel = np.abs(sc[nz] - ref[nz]) / ref[nz]
    _, _, _, (oi, orid, osc) = T.numpy_select(sc, valid, rank, cost, budget)
    around = osc[max(0, len(ci) - 2000):len(ci) + 2000].astype(np.float64)
    gaps = np.abs(np.diff(around))
    cut = float(csc[-1])
    flipped = np.array([[sc[i, r], ref[i, r]] for i, r in sym], dtype=np.float64).reshape(-1, 2)
    wr = wref.numpy()
    report = {
        "pictures": n_img, "regions": n_img * S, "budget_clicks": budget, "hip_round_seconds": dt,
        "selected_hip": len(ours), "selected_exact_c": int(len(ei)), "selected_reference_f32": len(theirs),
        "hip_equals_exact_c": eq,
        "vs_reference_f32_order": {
            "symmetric_difference": len(sym),
            "same_zero_regions": bool(np.array_equal(nz, sc != 0)),
            "max_abs_delta_score": float(np.abs(sc - ref).max()), "max_rel_delta_score": float(rel.max()),
            "median_rel_delta_score": float(np.median(rel)),
            "cutoff_score": cut,
            "median_gap_of_neighbouring_scores_at_cutoff": float(np.median(gaps)),
            "mean_gap_of_neighbouring_scores_at_cutoff": float(np.mean(gaps)),
            "regions_within_max_abs_delta_of_the_cutoff": int((np.abs(sc.astype(np.float64) - cut) <= float(np.abs(sc - ref).max())).sum()),
            "flipped_regions_max_distance_from_cutoff": float(np.abs(flipped - cut).max()) if len(sym) else 0.0,
            "flipped_regions_max_distance_from_cutoff_relative": float(np.abs(flipped - cut).max() / cut) if len(sym) else 0.0,
            "class_weight_max_rel_delta": float((np.abs(w_hip - wr) / wr).max()),
            "last_consumed_rank_equal": bool(len(ci) == len(ri)),
        },
        "host": {"threads": a.threads, "seconds_total": time.time() - t_all},
        "note": "HIP == oracle/exact.c means scores, class weights and the consumed prefix are bit-identical over the whole pool; against the "
                "reference's own f32 operation order the selected sets differ by symmetric_difference regions, every one of them within "
                "flipped_regions_max_distance_from_cutoff of the cut-off score (i.e. inside the rounding distance of two f32 evaluation orders)",
    }
    print(json.dumps(report), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(report, f, indent=1)
    if not (eq["class_weights"] and eq["scores"] and eq["consumed_prefix"]):
        sys.exit("HIP differs from oracle/exact.c")


if __name__ == "__main__":
    main()
