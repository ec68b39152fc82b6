#!/usr/bin/env python3
"""Golden summary for the web-Google-shaped surrogate (BASELINE configs[2] by shape; run in the build container):

    synth.webgraph_csr(916 428, seed 46): A*A by the REAL reference's omp_CSR_SpMM (oracle/_ref) -> nnz, structure hash,
    value checksums, P, per-bin row counts, 8-way flops partition; merged into tests/golden/golden_large.json under
    "web_surrogate_916428_46".  The totals the reference tree records for the real web-Google (tools/res.txt:1910) are
    kept next to them for comparison -- the surrogate matches the SHAPE (rows, ~5.5 entries per row, nnzC/P ~ 0.49), not
    the matrix: "surrogate; reference totals unpinned".

    make -C oracle ref && python tests/golden/make_golden_web.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import po, summarize, synth  # noqa: E402

assert po.have_ref(), "build oracle/_ref first: make -C oracle ref"


def main():
    t0 = time.time()
    m, seed = synth.WEB_M, 46
    rp, ci, v = synth.webgraph_csr(m, seed)
    A = po.CSRHost(rp, ci, v, m, m)
    Cm = po.ref_spmm(A, A, "omp")
    pref = po.ref_row_flops_prefix(A, A)
    flops = np.diff(pref)
    s = summarize(Cm)
    _, _, hv, hv_len = po.gpu_classify(flops)                    # hv of gpuFlopsClassify (incl. its dummy element)
    indeg = np.bincount(A.colInd, minlength=m)
    s.update({"m": m, "seed": seed, "nnzA": int(A.nnz), "P": int(pref[-1]), "max_row_flops": int(flops.max()),
              "max_out_degree": int(np.diff(A.rowPtr).max()), "max_in_degree": int(indeg.max()),
              "nnzC_over_P": round(Cm.nnz / float(pref[-1]), 4),
              "hv": [int(x) for x in hv], "hv_len": int(hv_len),
              "partition8": [int(x) for x in po.ref_equal_partition64(pref, 8)],
              "real_web_google_totals_res_txt_1910": {"N": 916428, "nnzA": 5105039, "nnzC": 29710164, "flops_2P": 121375672},
              "label": "surrogate; reference totals unpinned"})
    path = os.path.join(HERE, "golden_large.json")
    out = json.load(open(path))
    out["web_surrogate_916428_46"] = s
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("web surrogate", s["nnzA"], s["P"], s["nnz"], s["nnzC_over_P"], f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
