#!/usr/bin/env python3
"""LDS bank-conflict model of the owner-computes gathers on the benchmark set.

Host-only: rebuilds, for a sample of graph pairs, the LDS dword address every
lane gathers in every register slot of the static row-batch layouts (mgk_oc.h:
sorted row space, batches of 64 rows, grid walk of the first batch, running
walk of the others) and prices each gather instruction with the bank model of
MI355X_MICROARCH.md (ds_read_b32: two 32-lane groups, bank = dword address
mod 32, lanes on one address share the access; ds_read_b64 behaves the same
on 8-byte elements), for alternative layouts of p:

    python scripts/lds_sim.py [--pairs=3000] [--f64]

Prints LDS cycles per gather instruction (2.0 = conflict free) for
  ldp = n2 | 1 (the build's odd row stride), ldp = n2, ldp = 32,
and for lane orders of the sorted row space.
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np                                                   # noqa: E402
import cases                                                         # noqa: E402
from graphdot_amd.kernel.marginalized import _devicegraph as dgm      # noqa: E402
from graphdot_amd.kernel.marginalized._backend_hip import OC_STATIC_VARIANTS  # noqa: E402

DMAX = 4
ORDER = [(a, b) for p in range(DMAX * DMAX, -1, -1)
         for a in range(DMAX + 1) for b in range(DMAX + 1) if a * b == p]


def graph_arrays(dg):
    rp = np.asarray(dg.rowptr, dtype=np.int64)
    return rp, np.asarray(dg.nz)['j'].astype(np.int64)


def sorted_rows(rp1, rp2):
    """rowmap: (i1, i2) of every sorted position (mgk_oc.h rowmap phase)."""
    d1, d2 = np.diff(rp1), np.diff(rp2)
    n1, n2 = len(d1), len(d2)
    cnt1 = np.bincount(d1, minlength=DMAX + 1)
    cnt2 = np.bincount(d2, minlength=DMAX + 1)
    # nodes are stored by descending degree: class d starts after the higher ones
    st1 = np.array([cnt1[d + 1:].sum() for d in range(DMAX + 1)])
    st2 = np.array([cnt2[d + 1:].sum() for d in range(DMAX + 1)])
    off, o = {}, 0
    for a, b in ORDER:
        off[(a, b)] = o
        o += cnt1[a] * cnt2[b]
    i1, i2 = np.divmod(np.arange(n1 * n2), n2)
    a, b = d1[i1], d2[i2]
    base = np.array([off[(x, y)] for x, y in zip(a, b)])
    pos = base + (i1 - st1[a]) * cnt2[b] + (i2 - st2[b])
    rm = np.empty((n1 * n2, 2), dtype=np.int64)
    rm[pos, 0], rm[pos, 1] = i1, i2
    return rm, d1, d2


def fits(prod_sorted, L):
    N = len(prod_sorted)
    if N > 64 * len(L):
        return False
    return all(k * 64 >= N or prod_sorted[k * 64] <= L[k]
               for k in range(len(L)))


def slot_addresses(rp1, nj1, rp2, nj2, L, ldp, grid=True, lane_of=None,
                   swz=None):
    """dword index of p gathered by every lane in every slot: [S, 64].
    `swz(j1, j2)`: the column of element (j1, j2) within its row of p (round
    6: bank swizzles -- XOR of row bits into the column -- against the odd
    stride)."""
    if swz is None:
        swz = lambda j1, j2: j2
    rm, d1, d2 = sorted_rows(rp1, rp2)
    N = len(rm)
    S = sum(L)
    adr = np.zeros((S, 64), dtype=np.int64)
    s0 = 0
    for k, Lk in enumerate(L):
        for lane in range(64):
            pos = k * 64 + (lane if lane_of is None else lane_of[lane])
            if pos >= N:
                continue
            i1, i2 = rm[pos]
            a, b = d1[i1], d2[i2]
            n1_ = nj1[rp1[i1]:rp1[i1] + a]
            n2_ = nj2[rp2[i2]:rp2[i2] + b]
            if k == 0 and grid and Lk == DMAX * DMAX:
                for u in range(DMAX):
                    for v in range(DMAX):
                        if u < a and v < b:
                            adr[s0 + u * DMAX + v, lane] = \
                                n1_[u] * ldp + swz(n1_[u], n2_[v])
            else:
                for t in range(min(Lk, a * b)):
                    adr[s0 + t, lane] = n1_[t // b] * ldp + \
                        swz(n1_[t // b], n2_[t % b])
        s0 += Lk
    return adr


def gather_cycles(adr, banks=32):
    """LDS cycles of every gather instruction (rows of adr): per 32-lane
    group the largest number of distinct addresses on one bank."""
    cyc = np.zeros(len(adr), dtype=np.int64)
    for s, row in enumerate(adr):
        for half in (row[:32], row[32:]):
            u = np.unique(half)
            cyc[s] += np.bincount(u % banks, minlength=banks).max()
    return cyc


def main():
    n_pairs = 3000
    for a in sys.argv[1:]:
        if a.startswith('--pairs='):
            n_pairs = int(a.split('=')[1])
    G = cases.config3_graphs(1000)
    dgs = dgm.pack_many(G, np.float32)
    arrs = [graph_arrays(d) for d in dgs]
    rng = np.random.RandomState(0)
    layouts = [v.L for v in OC_STATIC_VARIANTS]
    tot = {}
    cnt = {}
    for _ in range(n_pairs):
        i, j = rng.randint(1000, size=2)
        rp1, nj1 = arrs[i]
        rp2, nj2 = arrs[j]
        if max(np.diff(rp1).max(), np.diff(rp2).max()) > DMAX:
            continue
        prod = np.sort(np.outer(np.diff(rp1), np.diff(rp2)).ravel())[::-1]
        L = next((L for L in layouts if fits(prod, L)), None)
        if L is None:
            continue
        n2 = len(rp2) - 1
        cases_ = {
            'odd stride, running walk': dict(ldp=n2 | 1, grid=False),
            'odd stride, grid': dict(ldp=n2 | 1, grid=True),
            'stride n2, grid': dict(ldp=n2, grid=True),
            'stride 32, grid': dict(ldp=32, grid=True),
            'stride 33, grid': dict(ldp=33, grid=True),
            # swizzles on rows of 32 cells (n2 <= 23 on this set): the row
            # index XORed into the column bits
            'stride 32, col ^ row': dict(
                ldp=32, grid=True, swz=lambda j1, j2: j2 ^ (j1 & 31)),
            'stride 32, col ^ 3 row': dict(
                ldp=32, grid=True, swz=lambda j1, j2: j2 ^ ((3 * j1) & 31)),
            'stride 32, col ^ 5 row': dict(
                ldp=32, grid=True, swz=lambda j1, j2: j2 ^ ((5 * j1) & 31)),
            'stride 32, col ^ 7 row': dict(
                ldp=32, grid=True, swz=lambda j1, j2: j2 ^ ((7 * j1) & 31)),
            'stride 32, col + 11 row (rotate)': dict(
                ldp=32, grid=True, swz=lambda j1, j2: (j2 + 11 * j1) & 31),
        }
        for name, kw in cases_.items():
            c = gather_cycles(slot_addresses(rp1, nj1, rp2, nj2, L, **kw))
            tot[name] = tot.get(name, 0) + c.sum()
            cnt[name] = cnt.get(name, 0) + len(c)
    for name in tot:
        print(f'{name:28s} {tot[name] / cnt[name]:.3f} LDS cycles per gather '
              f'({cnt[name]} gathers)')


if __name__ == '__main__':
    main()
