"""CPU: the C++ mirror's multi-threaded text loader (csrc/nlibs/COO.cc, readSNAPFile) reads what the reference's loader
reads (nlibs/COO.cc:48-158, through the oracle's restatement, itself pinned to the compiled reference in
test_oracle_vs_ref.py / the golden fixtures) -- for any thread count, on the reference's own fixture files and on
generated edge lists with the cases the format has: comments, a size line with 2 or 3 numbers, missing values, CRLF,
a line that is not an edge (reading stops there), fewer / more lines than declared, MatrixMarket 1-based and symmetric."""
import os
import subprocess

import numpy as np
import pytest

from helpers import DATA, ROOT, po

EXE = os.path.join(ROOT, "tests", "cpp", "parse_check.x")


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "parse_check.x"])


def mirror_read(path, isTrans, threads, tmp_path):
    out = tmp_path / f"dump_{threads}.bin"
    env = dict(os.environ, SMF_PARSE_THREADS=str(threads))
    r = subprocess.run([EXE, str(path), str(int(isTrans)), str(out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.int32)
    rows, cols, nnz = (int(x) for x in raw[:3])
    ri = raw[3:3 + nnz]
    ci = raw[3 + nnz:3 + 2 * nnz]
    v = raw[3 + 2 * nnz:3 + 3 * nnz].view(np.float32)
    return rows, cols, ri, ci, v, r.stdout


def check(path, isTrans, tmp_path, threads=(1, 3, 8)):
    rows, cols, ri, ci, v = po.read_snap(str(path), isTrans)
    for t in threads:
        g = mirror_read(path, isTrans, t, tmp_path)
        assert (g[0], g[1]) == (rows, cols), (path, t)
        assert np.array_equal(g[2], ri) and np.array_equal(g[3], ci), (path, t)
        assert np.array_equal(g[4].view(np.uint32), v.view(np.uint32)), (path, t)     # float bits
    return len(ri)


@pytest.mark.parametrize("name", ["test.mtx", "test2.mtx", "t2.snap", "tdata.snap", "own_graph.snap", "own_dups.mtx"])
@pytest.mark.parametrize("isTrans", [False, True])
def test_reference_fixture_files(name, isTrans, tmp_path):
    path = os.path.join(DATA, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not among the fixtures")
    check(path, isTrans, tmp_path)


def _edge_file(path, n, declared, rows, seed, header="# generated\n# more\n", three=False, values="some", crlf=False,
               bad_at=None, banner=None):
    rng = np.random.default_rng(seed)
    fr, to = rng.integers(0, rows, n), rng.integers(0, rows, n)
    val = rng.random(n).astype(np.float32)
    eol = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        if banner:
            f.write(banner + eol)
        f.write(header.replace("\n", eol))
        f.write((f"{rows} {rows} {declared}" if three else f"{rows} {declared}") + eol)
        for i in range(n):
            if bad_at is not None and i == bad_at:
                f.write("this is not an edge" + eol)
                continue
            if values == "all" or (values == "some" and i % 3 == 0):
                f.write(f"{fr[i]} {to[i]} {val[i]:.7g}{eol}")
            elif i % 7 == 0:
                f.write(f"  {fr[i]}\t{to[i]}  {eol}")                    # leading blanks, tab, trailing blanks
            else:
                f.write(f"{fr[i]} {to[i]}{eol}")


@pytest.mark.parametrize("case", ["plain", "three_numbers", "crlf", "bad_line", "short_file", "long_file", "no_final_newline"])
def test_generated_edge_lists(case, tmp_path):
    p = tmp_path / f"{case}.snap"
    n = 300000                                                            # ~3.5 MB: dozens of 64 KB chunks
    kw = dict(n=n, declared=n, rows=50000, seed=7)
    if case == "three_numbers":
        kw["three"] = True
    if case == "crlf":
        kw["crlf"] = True
    if case == "bad_line":
        kw["bad_at"] = 123457
    if case == "short_file":
        kw["declared"] = n + 1000                                         # fewer lines than declared
    if case == "long_file":
        kw["declared"] = n - 4321                                         # extra lines are never read
    _edge_file(p, **kw)
    if case == "no_final_newline":
        data = open(p, "rb").read().rstrip(b"\n")
        open(p, "wb").write(data)
    got = check(p, True, tmp_path, threads=(1, 2, 5, 16))
    want_n = {"bad_line": 123457, "long_file": n - 4321}.get(case, n)
    assert got == want_n


def test_matrix_market_general_and_symmetric(tmp_path):
    p = tmp_path / "g.mtx"
    _edge_file(p, n=5000, declared=5000, rows=400, seed=3, header="% comment\n", three=True, values="all",
               banner="%%MatrixMarket matrix coordinate real general")
    # 1-based indices: shift the generated 0-based ones by writing them +1
    lines = open(p).read().split("\n")
    body = [ln for ln in lines[3:] if ln]
    fixed = lines[:3] + [" ".join([str(int(a) + 1), str(int(b) + 1), c]) for a, b, c in (ln.split() for ln in body)]
    open(p, "w").write("\n".join(fixed) + "\n")
    check(p, False, tmp_path)
    q = tmp_path / "s.mtx"
    open(q, "w").write("\n".join(["%%MatrixMarket matrix coordinate real symmetric"] + fixed[1:]) + "\n")
    check(q, False, tmp_path)                                             # token-stream path, expands (i,j) -> (j,i)
