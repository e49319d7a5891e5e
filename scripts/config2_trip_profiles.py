"""Round 6, review item 3: could configuration 2 (256 Newman-Watts-Strogatz
graphs of 8-48 nodes, degree up to 8: reference
benchmark/kernel/marginalized/time_kernel.py:14-29) run on STATIC row-batch
layouts like the molecular set does?  A static layout fixes, at compile time,
how many slots each 64-row batch of a wave owns; the molecular set needs 11 of
them for 66 trip profiles.  This prints what configuration 2 would need: the
number of distinct per-wave trip profiles of its 32 896 pairs, how many pairs
the most common ones cover, and the padding a covering menu of k layouts costs
(greedy cover by dominance).  CPU only.
Usage: python scripts/config2_trip_profiles.py [--out profiles/r06_c2_trip_profiles.json]"""
import argparse
import collections
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases                                                  # noqa: E402
from graphdot_amd.kernel.marginalized._backend_hip import HIPBackend   # noqa: E402


def wave_profiles(h1, h2, W, D):
    """Trip profile of every wave of a pair dealt to W waves in snake order
    (mgk_oc.h row_pos): tuple of per-batch trip counts, per wave."""
    prods = np.outer(np.arange(D + 1), np.arange(D + 1))
    cnt = np.outer(h1[:D + 1], h2[:D + 1])
    order = np.argsort(-prods.ravel(), kind='stable')
    rows = np.repeat(prods.ravel()[order], cnt.ravel()[order])     # sorted degree products
    T = 64 * W
    R = -(-len(rows) // T)
    out = []
    for w in range(W):
        prof = []
        for k in range(R):
            chunk = (W - 1 - w) if (k & 1) else w
            pos = k * T + 64 * chunk
            prof.append(int(rows[pos]) if pos < len(rows) else 0)
        out.append(tuple(prof))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--dtype', default='f32')
    a = ap.parse_args()
    import cases as _c
    from graphdot_amd.kernel.marginalized import MarginalizedGraphKernel
    from graphdot_amd.kernel.marginalized._backend_hip import OCVariant
    graphs = _c.config2_graphs(256)
    knode, kedge, q = _c.config2b_kernels()
    backend = HIPBackend(real=np.float32 if a.dtype == 'f32' else np.float64)
    k = MarginalizedGraphKernel(knode, kedge, q=q, backend=backend)
    n = len(graphs)
    ii, jj = np.triu_indices(n)
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    jobs = np.column_stack((ii, jj)).astype(np.uint32).ravel().view(job_t)
    traits = k.traits(symmetric=True, eval_gradient=False)
    # the product's own assignment of every pair to a solver variant (host only)
    dgraphs, ek, C, fields = backend._graphs_and_kernels(graphs, knode, kedge, traits)
    arena = backend._host_arena(dgraphs, fields)
    part = backend._partition(dgraphs, jobs, C, backend._table_bytes(arena),
                              backend._global_tables(arena))
    _, used, order_all, launches = part
    hists = np.array([np.bincount(np.diff(np.asarray(g.rowptr, dtype=np.int64)), minlength=17)
                      for g in dgraphs])
    report = {}
    for L in launches:
        v = backend.variants[L['k']]
        if not isinstance(v, OCVariant) or v.S == 0 or v.L:
            continue
        members = order_all[L['offset']:L['offset'] + L['count']]
        prof_count = collections.Counter()
        slots = []
        for t in members:
            # a static layout is compile-time code: one layout for the whole
            # workgroup, which must dominate every wave's profile -- the
            # batch-wise maximum over the waves; the dynamic solver's time goes
            # with the LARGEST total over the waves (they meet at barriers)
            wp = wave_profiles(hists[ii[t]], hists[jj[t]], v.W, v.D)
            p = tuple(int(x) for x in np.max(np.array(wp), axis=0))
            prof_count[p] += 1
            slots.append(max(sum(q_) for q_ in wp))
        total = sum(prof_count.values())
        common = prof_count.most_common()
        cover = np.cumsum([c for _, c in common]) / total
        profs = np.array([np.pad(p, (0, 16 - len(p))) for p, _ in common])
        weights = np.array([c for _, c in common], float)
        served = np.zeros(len(profs), bool)
        menu, padding = [], 0.0
        for _ in range(12):
            if served.all():
                break
            best, best_gain = None, -1
            for cand in profs[~served][:300]:
                dom = (profs <= cand).all(axis=1) & ~served
                gain = weights[dom].sum() / max(cand.sum(), 1)
                if gain > best_gain:
                    best, best_gain = cand, gain
            dom = (profs <= best).all(axis=1) & ~served
            padding += float((weights[dom] * (best.sum() - profs[dom].sum(axis=1))).sum())
            served |= dom
            menu.append([int(x) for x in best if x])
        name = f'W{v.W}_S{v.S}_R{v.R}'
        report[name] = dict(
            pairs=int(L['count']), workgroup_profiles=total, distinct_profiles=len(common),
            top12_cover=float(cover[min(11, len(cover) - 1)]),
            mean_slots=float(np.mean(slots)),
            menu_of_12_serves=float(weights[served].sum() / total),
            menu_of_12_padding=float(padding / max(weights[served].sum(), 1)),
            menu=menu, most_common=[(list(p), c) for p, c in common[:6]])
        static_slots = float((weights * profs.sum(axis=1)).sum() / total)
        print(f'{name}: {L["count"]} pairs, {len(common)} distinct workgroup profiles (batch-wise maximum over '
              f'the waves); the 12 most common cover {cover[min(11, len(cover) - 1)]:.1%}; slots on the '
              f'critical wave: dynamic {np.mean(slots):.1f}, one exact static layout per profile '
              f'{static_slots:.1f} (+{static_slots / np.mean(slots) - 1:.0%}), a greedy menu of 12 '
              f'dominating layouts {static_slots + padding / max(weights[served].sum(), 1):.1f} '
              f'(+{(static_slots + padding / max(weights[served].sum(), 1)) / np.mean(slots) - 1:.0%})')
        report[name]['slots_dynamic_critical_wave'] = float(np.mean(slots))
        report[name]['slots_static_exact'] = static_slots
        report[name]['slots_static_menu_of_12'] = static_slots + padding / max(weights[served].sum(), 1)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
