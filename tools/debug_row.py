#!/usr/bin/env python3
"""GPU debugging aid: symbolic count of single rows (and sub-selections of their A entries) against numpy."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_matrix_with_flops_amd import synth, hipspgemm as hs

def sym_count(h, rpA, ciA, dB, nB, nnzB):
    m = len(rpA) - 1
    dIA, dJA = hs.h2d(rpA.astype(np.int32)), hs.h2d(ciA.astype(np.int32))
    dIC = hs.dev_alloc(4 * (m + 1))
    hs.spgemm_symbolic_raw(h, dIA, dJA, len(ciA), dB[0], dB[1], nnzB, m, nB, nB, dIC)
    ic = hs.d2h(dIC, m + 1, np.int32)
    for p in (dIA, dJA, dIC): hs.dev_free(p)
    return np.diff(ic)

def main():
    m, seed, row = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rp, ci, v = synth.powerlaw_csr(m, seed, 2)
    deg = np.diff(rp).astype(np.int64)
    h = hs.Handle(0)
    dB = (hs.h2d(rp.astype(np.int32)), hs.h2d(ci.astype(np.int32)))
    cols = ci[rp[row]:rp[row + 1]]
    bl = deg[cols]
    def want(sel):
        s = set()
        for j in sel: s.update(ci[rp[j]:rp[j + 1]].tolist())
        return len(s)
    def run(sel, tag):
        sel = np.asarray(sel)
        got = [int(sym_count(h, np.array([0, len(sel)]), sel, dB, m, len(ci))[0]) for _ in range(4)]
        print(f"{tag}: entries={len(sel)} flops={int(deg[sel].sum())} want={want(sel)} got={got}")
    run(cols, "full row")
    run(cols[bl < 64], "short only")
    run(cols[bl >= 64], "long only")
    for g in range(0, len(cols), 64):
        run(cols[g:g + 64], f"group {g//64}")
        c = cols[g:g + 64]; b = bl[g:g + 64]
        run(c[b < 64], f"group {g//64} short")
        run(c[b >= 64], f"group {g//64} long")

if __name__ == "__main__":
    main()
