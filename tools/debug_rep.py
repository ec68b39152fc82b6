#!/usr/bin/env python3
"""GPU debugging aid: repeat the symbolic phase and list rows whose count differs from the oracle each time."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_matrix_with_flops_amd import synth, hipspgemm as hs
from oracle import pyoracle as po
m, seed, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
only_big = len(sys.argv) > 4 and sys.argv[4] == "big"
rp, ci, v = synth.powerlaw_csr(m, seed, 2)
A = po.CSRHost(rp, ci, v, m, m)
wcnt = np.diff(po.omp_spmm(A, A).rowPtr)
deg = np.diff(rp).astype(np.int64)
flops = np.bincount(np.repeat(np.arange(m), deg), weights=deg[ci], minlength=m).astype(np.int64)
h = hs.Handle(0)
dIB, dJB = hs.h2d(rp.astype(np.int32)), hs.h2d(ci.astype(np.int32))
rpA, ciA = rp, ci
if only_big:
    keep = flops > 4096
    d2 = np.where(keep, deg, 0)
    rpA = np.zeros(m + 1, dtype=np.int64); np.cumsum(d2, out=rpA[1:])
    ciA = ci[np.repeat(keep, deg)]
    wcnt = np.where(keep, wcnt, 0)
dIA, dJA = hs.h2d(rpA.astype(np.int32)), hs.h2d(ciA.astype(np.int32))
dIC = hs.dev_alloc(4 * (m + 1))
for it in range(reps):
    hs.spgemm_symbolic_raw(h, dIA, dJA, len(ciA), dIB, dJB, len(ci), m, m, m, dIC)
    g = np.diff(hs.d2h(dIC, m + 1, np.int32))
    bad = np.nonzero(g != wcnt)[0]
    print(it, "bad rows:", [(int(r), int(flops[r]), int(g[r]), int(wcnt[r])) for r in bad[:8]], len(bad))
