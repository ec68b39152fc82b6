#!/usr/bin/env python3
"""Generate tests/golden/*.npz / golden.json from the REAL reference (oracle/_ref/libref.so).

Run in the build container only (needs /root/reference to have built _ref):
    make -C oracle ref && python tests/golden/make_golden.py
The outputs are plain data (inputs + expected outputs); no reference source is stored.

What is recorded
  fixtures.npz   for every file under tests/golden/data: the CSR the reference loads
                 (readSNAPFile(f,false)+orderedAndDuplicatesRemoving+toCSR, and readSNAPFile(f,true/false)+rmclInit
                 where the matrix is square), A*A by sequential_CSR_SpMM in the reference's own first-touch
                 order, and R-MCL results after 1..3 iterations (RMCL(...,SEQ)).
  synth_small.npz  full A*A for small synthetic instances + random rectangular A*B pairs (reference output).
  golden.json    summaries (nnzC, structure hash, value checksums, P, bins) for larger synthetic instances
                 computed with the reference's omp_CSR_SpMM, plus the known answers quoted in SURVEY.md §4.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import DATA, po, random_csr, summarize, synth_csr  # noqa: E402

assert po.have_ref(), "build oracle/_ref first: make -C oracle ref"


def pack(prefix, C, out):
    out[prefix + "_rowPtr"] = C.rowPtr
    out[prefix + "_colInd"] = C.colInd
    out[prefix + "_values"] = C.values
    out[prefix + "_shape"] = np.array([C.rows, C.cols], dtype=np.int64)


def main():
    fx = {}
    meta = {"fixtures": {}, "synth": {}, "survey_known_answers": {}}
    for name in sorted(os.listdir(DATA)):
        path = os.path.join(DATA, name)
        key = name.replace(".", "_")
        if name == "tdata.snap":
            # no "rows nnz" line: the reference parses the first edge "0 0" as rows=0 nnz=0 (COO.cc:80-89) and
            # then reads v[0] of an empty vector in orderedAndDuplicatesRemoving (COO.cc:245-258) -> crashes here.
            # Kept only as the "empty input" edge case for our own loader.
            meta["fixtures"][name] = {"rows": 0, "cols": 0, "nnz": 0, "note": "reference UB/segfault; expect empty"}
            continue
        A = po.ref_load(path, isTrans=False, mode=0)
        pack(key + "_load", A, fx)
        info = {"rows": A.rows, "cols": A.cols, "nnz": A.nnz}
        if A.rows == A.cols and A.rows > 0:
            Cm = po.ref_spmm(A, A, "sequential")
            pack(key + "_AA", Cm, fx)
            info["nnzC"] = Cm.nnz
            for trans in (False, True):
                M = po.ref_load(path, isTrans=trans, mode=1)
                pack(f"{key}_init_t{int(trans)}", M, fx)
            for it in (1, 2, 3):
                R = po.ref_rmcl(path, it, 0)
                pack(f"{key}_rmcl{it}", R, fx)
        meta["fixtures"][name] = info
    np.savez_compressed(os.path.join(HERE, "fixtures.npz"), **fx)

    sm = {}
    for (m, seed, base) in [(64, 3, 2), (512, 7, 2), (1024, 9, 4)]:
        A = synth_csr(m, seed, base)
        Cm = po.ref_spmm(A, A, "sequential")
        pack(f"synth_{m}_{seed}_{base}_AA", Cm, sm)
    for idx, (r, k, c, d, seed) in enumerate([(37, 53, 41, 0.12, 1), (128, 96, 200, 0.05, 2), (5, 300, 7, 0.3, 3)]):
        A = random_csr(r, k, d, seed, sorted_rows=False)
        B = random_csr(k, c, d, seed + 100, sorted_rows=False)
        pack(f"rect{idx}_A", A, sm)
        pack(f"rect{idx}_B", B, sm)
        pack(f"rect{idx}_C", po.ref_spmm(A, B, "sequential"), sm)
    np.savez_compressed(os.path.join(HERE, "synth_small.npz"), **sm)

    for (m, seed, base) in [(4096, 11, 2), (32768, 13, 2), (65536, 17, 4), (262144, 42, 2), (1048576, 43, 2)]:
        A = synth_csr(m, seed, base)
        Cm = po.ref_spmm(A, A, "omp")
        pref = po.ref_row_flops_prefix(A, A)
        flops = np.diff(pref)
        rf, groups, tops = po.ref_group_bins(A, A)
        s = summarize(Cm)
        s.update({"m": m, "seed": seed, "base": base, "nnzA": A.nnz, "P": int(pref[-1]),
                  "max_row_flops": int(flops.max()), "group_tops": [int(x) for x in tops],
                  "partition8": [int(x) for x in po.ref_equal_partition64(pref, 8)]})
        meta["synth"][f"{m}_{seed}_{base}"] = s
        print(m, seed, base, s["nnz"], s["P"])

    # numbers the reference tree / SURVEY.md §4 already hold (not produced by this script)
    meta["survey_known_answers"] = {
        "test2_mtx_AA_rowPtr": [0, 3, 6, 8, 11],
        "test2_mtx_AA_row0": [[0, 15.02], [2, 1.68], [3, -26.672001]],
        "test2_mtx_AA_row1": [[1, -3.18], [3, 65.192398], [0, -75.585999]],
        "test2_mtx_AA_row2": [[1, 2.76], [3, 4.6]],
        "test2_mtx_AA_row3": [[1, 10.920001], [3, 84.45961], [0, -74.074005]],
        "t2_snap_rmcl3": [[0, 0, 1.0], [1, 0, 1.0], [2, 2, 1.0]],
        "synth_262144_42_2": {"nnzA": 3887048, "P": 58865303, "nnzC": 55418390},
        "res_txt_web_google": {"N": 916428, "nnzA": 5105039, "nnzC": 29710164, "flops": 121375672},
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
