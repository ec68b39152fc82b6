"""GPU (-m gpu): the R-MCL caller of the hot path — device prune step + gpuRmclIter — against the
oracle's seqRmclIter restatement and the golden R-MCL results made from the real reference; plus the C++ mirror
driven by the reference-style test program tests/cpp/testGpuSpMM.cc."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import DATA, GOLDEN, ROOT, assert_rmcl_step, canonical_arrays, po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_parity import FX, SQUARE, to_hs, unpack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert hs.device_count() >= 1


def ordered(M):
    c, v = canonical_arrays(M.rowPtr, M.colInd, M.values)
    return np.asarray(M.rowPtr), c, v


@pytest.mark.parametrize("name", SQUARE)
@pytest.mark.parametrize("iters", [1, 3])
def test_rmcl_fixtures_match_reference_goldens(name, iters):
    Mt = po.load(os.path.join(DATA, name), isTrans=True, mode=1)          # RMCL() reads the transpose
    got = hs.gpuRmclIter(iters, to_hs(Mt), to_hs(Mt))
    want = unpack(FX, f"{name.replace('.', '_')}_rmcl{iters}")            # RMCL(file, iters, SEQ) of the reference
    gr, gc, gv = ordered(got)
    wr, wc, wv = ordered(want)
    assert np.array_equal(gr, wr) and np.array_equal(gc, wc)
    assert np.allclose(gv, wv, rtol=1e-6, atol=0.0)                       # north_star: values within 1e-6 relative


def _graph(m, seed):
    A = synth_csr(m, seed, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    return po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))     # transpose + self loops + row-normalise


def _device_steps(Mt, iters, fused=True):
    """The device-resident loop one step at a time: yields (k, Mt_k, Mt_k+1).  fused: hip_rmcl_expand_prune per step
    (what hip_gpuRmclIter and ShardedRMCL run); otherwise hip_gpuSpMM + hip_rmcl_prune."""
    import torch
    from sparse_matrix_with_flops_amd.dist import HipEngine, make_matrix
    eng = HipEngine(0)
    Mg = make_matrix(eng, Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
    cur_dev, cur_host = Mg, Mt
    for k in range(iters):
        rp, ci, v = eng.expand_prune(Mg, cur_dev, fused=fused)
        torch.cuda.synchronize()
        nxt_host = po.CSRHost(rp.cpu().numpy(), ci.cpu().numpy(), v.cpu().numpy(), Mt.rows, Mt.cols)
        yield k, cur_host, nxt_host
        cur_dev = {"rowPtr": rp, "colInd": ci, "values": v, "rows": Mt.rows, "cols": Mt.cols, "nnz": int(ci.numel())}
        cur_host = nxt_host


def _ragged(rows, cols, lens, seed, pools=None):
    """CSR whose row i has lens[i] distinct random columns (unsorted) and values in [0.25, 1.25); pools: {row: array of
    the columns that row draws from}."""
    rng = np.random.default_rng(seed)
    rp = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=rp[1:])
    ci = np.empty(int(rp[-1]), dtype=np.int32)
    for i in range(rows):
        ci[rp[i]:rp[i + 1]] = rng.choice(cols if not pools or i not in pools else pools[i], size=int(lens[i]), replace=False)
    v = (rng.random(len(ci)) + 0.25).astype(np.float32)
    return po.CSRHost(rp, ci, v, rows, cols)


@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("n,big", [(6000, True), (300000, True), (6000, False), (300000, False)])
def test_fused_expand_prune_every_kernel(n, big, sym, monkeypatch):
    """hip_rmcl_expand_prune on a rectangular product whose rows land in EVERY bin, with duplicate-free rows (list
    staging) and colliding rows (hash tables) in each: one step against the oracle.
    big: rows beyond 4096 products exist (n = 300000 sends them through the hash kernel and its multi-pass rows,
    n = 6000 through the rank kernel): they alone get a symbolic kernel and are fixed up in place after the numeric one;
    all other rows are laid out by their product counts, no symbolic pass.  sym: SPGEMM_RMCL_SYMBOLIC forces the
    variant with the full symbolic pass (exact row pointers, duplicate-free rows staged as plain lists)."""
    import torch
    if sym:
        monkeypatch.setenv("SPGEMM_RMCL_SYMBOLIC", "1")
    else:
        monkeypatch.delenv("SPGEMM_RMCL_SYMBOLIC", raising=False)
    from sparse_matrix_with_flops_amd.dist import HipEngine, make_matrix
    rng = np.random.default_rng(5)
    m, k = 2500, 4000
    pools = {}
    if big:
        blen = rng.choice([0, 1, 2, 3, 5, 9, 20, 70, 200, 900], size=k, p=[.05, .2, .2, .15, .15, .1, .08, .04, .02, .01])
        alen = rng.choice([0, 1, 2, 4, 8, 20, 60, 150, 400, 1500], size=m, p=[.03, .1, .15, .2, .2, .15, .1, .04, .02, .01])
    else:
        blen = rng.choice([0, 1, 2, 3, 5, 9, 20, 66], size=k, p=[.05, .2, .2, .15, .15, .1, .1, .05])
        alen = rng.choice([0, 1, 2, 4, 8, 20, 40, 60], size=m, p=[.03, .1, .15, .2, .2, .15, .1, .07])
        longB = np.nonzero(blen == 66)[0]
        for i in range(0, m, 50):                                         # rows of 45..60 long B rows: 2970..3960 products
            alen[i] = 45 + (i // 50) % 16
            pools[i] = longB
    A = _ragged(m, k, alen, 11, pools)
    B = _ragged(k, n, blen, 12)
    eng = HipEngine(0)
    dA = make_matrix(eng, A.rowPtr, A.colInd, A.values, A.rows, A.cols)
    dB = make_matrix(eng, B.rowPtr, B.colInd, B.values, B.rows, B.cols)
    rp, ci, v = eng.expand_prune(dA, dB, fused=True)
    torch.cuda.synchronize()
    st = eng.stats()
    assert all(r > 0 for r in st["bin_rows"][:8]) and (st["bin_rows"][8] > 0) == big, st["bin_rows"]
    assert (st["nnzC"] == -1) == (not sym)                                # -1: nnz of the product not computed
    got = po.CSRHost(rp.cpu().numpy(), ci.cpu().numpy(), v.cpu().numpy(), m, n)
    # every row at the 3e-6 of the step; rows of more than 512 product entries against the float64 evaluation of the row
    # rule (the oracle's sequential float32 kept sum is itself off by up to n*eps/2: helpers.assert_rmcl_step)
    ndiff, ties, want = assert_rmcl_step(got, A, B, what=f"fused step, n={n}")
    print(f"n={n}: bins {st['bin_rows']}, {ndiff} rows differ ({ties} tie rows), nnz {got.nnz} vs {want.nnz}")
    # and the two-step path agrees with the same oracle on the same inputs
    rp2, ci2, v2 = eng.expand_prune(dA, dB, fused=False)
    torch.cuda.synchronize()
    got2 = po.CSRHost(rp2.cpu().numpy(), ci2.cpu().numpy(), v2.cpu().numpy(), m, n)
    assert_rmcl_step(got2, A, B, what=f"two-step, n={n}")


def test_fused_and_two_step_loops_agree(monkeypatch):
    """Three iterations of the loop three ways on the same graph -- fused without the symbolic pass (what the loop runs:
    no row of this graph passes 4096 products), fused with it (SPGEMM_RMCL_SYMBOLIC), and hip_gpuSpMM + hip_rmcl_prune --
    every step of each against the oracle."""
    Mt = _graph(20000, 91)
    for fused, sym in ((True, False), (True, True), (False, False)):
        if sym:
            monkeypatch.setenv("SPGEMM_RMCL_SYMBOLIC", "1")
        else:
            monkeypatch.delenv("SPGEMM_RMCL_SYMBOLIC", raising=False)
        for k, cur, nxt in _device_steps(Mt, 3, fused=fused):
            assert_rmcl_step(nxt, Mt, cur, what=f"fused={fused} symbolic={sym} iteration {k + 1}")


@pytest.mark.parametrize("hubs", [False, True])
def test_device_loop_leaves_Mt_unpacked_between_iterations(hubs, monkeypatch):
    """hip_gpuRmclIter_device: between iterations Mt stays where the epilogues wrote it ({start, kept} per row, read by
    k_row_flops' IBlen) and only the last iteration packs.  k iterations in one call against the oracle's step from the
    (k-1)-iteration result of another call; hubs: rows beyond 4096 products (symbolic kernel + in-place fix-up of their
    scratch rows) take part.  SPGEMM_RMCL_PACK (pack after every iteration) gives the same matrices; zero iterations
    return a copy of Mt."""
    monkeypatch.delenv("SPGEMM_RMCL_PACK", raising=False)
    monkeypatch.delenv("SPGEMM_RMCL_SYMBOLIC", raising=False)
    if hubs:
        rng = np.random.default_rng(8)
        m = 6000
        lens = rng.choice([0, 1, 2, 4, 8, 20, 60, 150, 400, 1500], size=m, p=[.03, .1, .15, .2, .2, .15, .1, .04, .02, .01])
        M0 = _ragged(m, m, lens, 21)
    else:
        M0 = _graph(20000, 77)
        m = M0.rows
    h = hs.Handle(0)
    dM = to_hs(M0).toGpuCSR()

    def run(k):
        d = hs.gpuRmclIter_device(k, dM, dM, h)
        st = h.stats()
        out = d.toCpuCSR()
        d.deviceDispose()
        return po.CSRHost(out.rowPtr, out.colInd, out.values, m, m), st

    prev = M0
    for k in (1, 2, 3):
        cur, st = run(k)
        if hubs and k == 1:
            assert st["bin_rows"][8] > 0, st["bin_rows"]
        # k > 1: `prev` comes from ANOTHER run than the state this call multiplied.  Two runs sum a long row in different
        # orders (LDS atomics of several waves), so an entry of prev that sits on its row's threshold may be kept in one
        # and dropped in the other; on the hub graph such an entry is ~1e-5 of its row's kept sum, which renormalises
        # the row by that much and moves every threshold it feeds.  Hence the wider bounds there (a wrong row start or
        # length -- what this test is after -- is off by orders of magnitude more).
        loose = dict(rel=5e-5, tie_rel=5e-5) if hubs and k > 1 else {}
        ndiff, ties, _ = assert_rmcl_step(cur, M0, prev, what=f"device loop, {k} iterations (hubs={hubs})", **loose)
        prev = cur
    monkeypatch.setenv("SPGEMM_RMCL_PACK", "1")
    packed, _ = run(3)
    monkeypatch.delenv("SPGEMM_RMCL_PACK")
    two, _ = run(2)
    assert_rmcl_step(packed, M0, two, what="device loop packing every iteration", **(dict(rel=5e-5, tie_rel=5e-5) if hubs else {}))
    zero, _ = run(0)
    assert np.array_equal(zero.rowPtr, M0.rowPtr) and np.array_equal(zero.colInd, M0.colInd) and np.array_equal(zero.values, M0.values)
    # the inputs were not touched
    back = dM.toCpuCSR()
    assert np.array_equal(back.rowPtr, M0.rowPtr) and np.array_equal(back.colInd, M0.colInd) and np.array_equal(back.values, M0.values)
    dM.deviceDispose()
    h.close()


def test_rmcl_synthetic_graph_three_iterations():
    """Power-law graph, 3 iterations, every step checked against the oracle from the device's own previous state: rows
    are identical (kept columns bit-exact, values 1e-6) except where an entry sits within 4 float32 ulps of the prune
    threshold -- the device sums a row in another order than the sequential CPU loop.  The number of such rows is
    counted and bounded by the number of rows that HAVE such an entry."""
    Mt = _graph(20000, 91)
    for k, cur, nxt in _device_steps(Mt, 3):
        ndiff, ties, want = assert_rmcl_step(nxt, Mt, cur, what=f"iteration {k + 1}")
        print(f"iteration {k + 1}: {ndiff} rows differ from the oracle, {ties} rows hold a threshold tie, nnz {nxt.nnz} vs {want.nnz}")
        gl = np.diff(nxt.rowPtr)
        rs = np.add.reduceat(nxt.values.astype(np.float64), nxt.rowPtr[:-1][gl > 0])
        assert np.allclose(rs, 1.0, atol=1e-5)                            # every row is a distribution again


def test_config4_rmcl_500k_nodes_ten_iterations():
    """BASELINE configs[4] on one GPU: 500 000-node power-law graph (seed 45), 10 iterations of the device-resident
    loop.  Per iteration: (1) the step from the device's own state equals the oracle's step up to counted threshold
    ties (assert_rmcl_step), (2) rows sum to 1, (3) nnz follows the per-iteration summary in golden_large.json (made
    with the reference-pinned kernels and cross-checked against RMCL(file, 10, OMP) of the real reference) within the
    drift the ties allow."""
    G = json.load(open(os.path.join(GOLDEN, "golden_large.json")))["rmcl_500000_45"]
    Mt = _graph(G["m"], G["seed"])
    assert Mt.nnz == G["nnz0"]
    total_ties = 0
    for k, cur, nxt in _device_steps(Mt, G["iters"]):
        ndiff, ties, want = assert_rmcl_step(nxt, Mt, cur, what=f"iteration {k + 1}")
        g = G["per_iter"][k]
        total_ties += ndiff
        gl = np.diff(nxt.rowPtr)
        rs = np.add.reduceat(nxt.values.astype(np.float64), nxt.rowPtr[:-1][gl > 0])
        print(f"iteration {k + 1}: nnz {nxt.nnz} (golden {g['nnz']}), {ndiff} of {ties} tie rows differ from the oracle step")
        assert np.allclose(rs, 1.0, atol=1e-5)
        # a flipped tie changes later iterations a little: nnz stays within 0.05 % + the ties seen so far
        assert abs(int(nxt.nnz) - g["nnz"]) <= 5e-4 * g["nnz"] + 64 * (total_ties + 1)


def test_sharded_rmcl_single_rank_steps_match_the_oracle():
    """dist.ShardedRMCL at world size 1 (device-resident loop on torch tensors as the replicated Mt): every iteration is
    checked from the loop's own previous state with assert_rmcl_step -- identical rows at 3e-6 except counted threshold
    ties -- and the final state against hip_gpuRmclIter up to the ties seen on the way."""
    import torch
    from sparse_matrix_with_flops_amd.dist import HipEngine, ShardedRMCL
    A = synth_csr(12000, 57, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    Mt = po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))
    host = (Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
    job = ShardedRMCL(HipEngine(0), host, host)
    cur, flipped = Mt, 0
    for k in range(3):
        job.iterate(1)
        rp, ci, v = job.result_host()
        torch.cuda.synchronize()
        nxt = po.CSRHost(rp, ci, v, Mt.rows, Mt.cols)
        ndiff, ties, _ = assert_rmcl_step(nxt, Mt, cur, what=f"sharded loop, world 1, iteration {k + 1}")
        flipped += ndiff
        cur = nxt
    want = hs.gpuRmclIter(3, to_hs(Mt), to_hs(Mt))
    if flipped == 0:                                  # no tie flipped on the way: the two loops hold the same matrix
        wr, wc, wv = ordered(want)
        gr, gc, gv = ordered(cur)
        assert np.array_equal(gr, wr) and np.array_equal(gc, wc) and np.allclose(gv, wv, rtol=1e-5, atol=0.0)
    else:
        assert abs(cur.nnz - want.nnz) <= 64 * flipped


def test_cpp_mirror_runs_the_reference_test_protocol():
    """tests/cpp/testGpuSpMM.cc == tests/testGpuSpMM.cc + nrmcl.cc of the reference, built on the C++ mirror."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "tests", "cpp")])
    exe = os.path.join(ROOT, "tests", "cpp", "testGpuSpMM.x")
    for name, extra in (("test2.mtx", []), ("own_graph.snap", ["--rmcl", "3"]), ("t2.snap", ["--rmcl", "3"]),
                        ("own_dups.mtx", [])):
        out = subprocess.run([exe, os.path.join(DATA, name)] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "Differs" not in out.stdout and "Same" in out.stdout
    # nrmcl with the reference's flags (process_args.cc:12-27): --input/-i --maxIters/-m --stride --rmclOptions/-r
    exe = os.path.join(ROOT, "tests", "cpp", "nrmcl.x")
    # ... and --shared None|Shared|L1 (process_args.cc:17,60-63; nGpuSpMM.cc:297-299): a reference command line runs unchanged
    for args in (["--input", os.path.join(DATA, "t2.snap"), "--maxIters", "3", "--stride", "128", "--rmclOptions", "GPU"],
                 ["-i", os.path.join(DATA, "own_graph.snap"), "-m", "2", "-r", "GPU", "--stats"],
                 ["--input", os.path.join(DATA, "t2.snap"), "--maxIters", "3", "--shared", "L1", "--rmclOptions", "GPU"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "rmclOption= GPU" in out.stdout and "Same" in out.stdout and "Diffs" not in out.stdout
        assert "unrecognized option" not in out.stderr
        assert ("SharedOption= CachePreferL1" if "--shared" in args else "SharedOption= CachePreferNone") in out.stdout
        if "--stats" in args:
            assert "Total sum =" in out.stdout
    # --stats also writes the reference's per-iteration drift report, percent.stats (nlibs/qrmcl.cc:17-24,65-70): one line
    # per iteration with CSR::differsStats(Mt -> new Mt); recomputed here from the oracle's iterations (tie-free fixture)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = subprocess.run([exe, "-i", os.path.join(DATA, "own_graph.snap"), "-m", "3", "-r", "GPU", "--stats"],
                             capture_output=True, text=True, timeout=300, cwd=td)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = open(os.path.join(td, "percent.stats")).read().splitlines()
    Mt = po.load(os.path.join(DATA, "own_graph.snap"), isTrans=True, mode=1)
    assert lines[0] == f"rows {Mt.rows}" and lines[1].startswith("percent\t") and len(lines) == 2 + 3
    percents = np.array([float(x) for x in lines[1].split("\t")[1].split()], dtype=np.float32)
    assert list(percents) == [-30, -20, -5, 0, 5, 20, 30, 100]
    cur = Mt
    for it in range(3):
        nxt = po.rmcl_iters(Mt, cur, 1)
        a, b = np.diff(cur.rowPtr).astype(np.int64), np.diff(nxt.rowPtr).astype(np.int64)
        counts = np.zeros(len(percents) + 4, dtype=np.int64)
        for ai, bi in zip(a, b):
            if ai == 0:
                counts[len(percents) + (1 if bi > 0 else 2)] += 1
            elif ai == bi:
                counts[len(percents) + 3] += 1
            else:
                ch = np.float32(bi - ai) / np.float32(ai)
                k = int(np.argmax(ch < percents)) if np.any(ch < percents) else len(percents)
                counts[k] += 1
        head, body = lines[2 + it].split(":\t")
        assert head.strip() == str(it) and [int(x) for x in body.split()] == counts.tolist(), (it, lines[2 + it], counts)
        assert counts.sum() == Mt.rows
        cur = nxt


def test_fused_expand_prune_degenerate_shapes():
    """hip_rmcl_expand_prune on products with nothing in them: no rows, rows without products (B empty, A empty) and a
    single 1x1 product -- the paths that bypass the fused kernels."""
    import torch
    from sparse_matrix_with_flops_amd.dist import HipEngine, make_matrix
    eng = HipEngine(0)

    def mk(rp, ci, v, r, c):
        return make_matrix(eng, np.asarray(rp, np.int32), np.asarray(ci, np.int32), np.asarray(v, np.float32), r, c)

    # A 3x4 with entries, B 4x5 empty: every row of the product is empty
    A = mk([0, 2, 2, 3], [0, 3, 1], [1.0, 2.0, 3.0], 3, 4)
    B = mk([0, 0, 0, 0, 0], [], [], 4, 5)
    rp, ci, v = eng.expand_prune(A, B)
    torch.cuda.synchronize()
    assert rp.cpu().tolist() == [0, 0, 0, 0] and ci.numel() == 0
    # A empty
    A0 = mk([0, 0, 0, 0], [], [], 3, 4)
    B1 = mk([0, 1, 2, 2, 3], [0, 4, 2], [1.0, 1.0, 1.0], 4, 5)
    rp, ci, v = eng.expand_prune(A0, B1)
    torch.cuda.synchronize()
    assert rp.cpu().tolist() == [0, 0, 0, 0] and ci.numel() == 0
    # 1x1
    A1 = mk([0, 1], [0], [0.5], 1, 1)
    rp, ci, v = eng.expand_prune(A1, A1)
    torch.cuda.synchronize()
    assert rp.cpu().tolist() == [0, 1] and ci.cpu().tolist() == [0] and abs(float(v.cpu()[0]) - 1.0) < 1e-6
    # one row keeps exactly what the rule says: values (3, 1, 1) -> squares 9, 1, 1: avg 11/3, max 9 > avg -> th clamps low
    A2 = mk([0, 3], [0, 1, 2], [3.0, 1.0, 1.0], 1, 3)
    I3 = mk([0, 1, 2, 3], [0, 1, 2], [1.0, 1.0, 1.0], 3, 3)
    rp, ci, v = eng.expand_prune(A2, I3)
    torch.cuda.synchronize()
    got = po.CSRHost(rp.cpu().numpy(), ci.cpu().numpy(), v.cpu().numpy(), 1, 3)
    Ah = po.CSRHost(np.array([0, 3], np.int32), np.array([0, 1, 2], np.int32), np.array([3, 1, 1], np.float32), 1, 3)
    Ih = po.CSRHost(np.array([0, 1, 2, 3], np.int32), np.array([0, 1, 2], np.int32), np.ones(3, np.float32), 3, 3)
    assert_rmcl_step(got, Ah, Ih, what="1x3 row")


def test_device_loop_gives_up_on_the_fused_step_behind_an_unpacked_Mt(monkeypatch):
    """Iteration 1 runs fused and leaves Mt unpacked; iteration 2 has more products than the scratch bound allows
    (SPGEMM_RMCL_MAXP, a test hook for the 2^30 bound) and runs as SpGEMM + prune: the unpacked Mt is packed first
    (rmcl_pack_rows).  Iteration 3 is above the bound too and stays on that path, now with a packed Mt."""
    monkeypatch.delenv("SPGEMM_RMCL_PACK", raising=False)
    M0 = _graph(20000, 77)
    m = M0.rows
    dM = to_hs(M0).toGpuCSR()
    h = hs.Handle(0)

    def run(k):
        d = hs.gpuRmclIter_device(k, dM, dM, h)
        out = d.toCpuCSR()
        d.deviceDispose()
        return po.CSRHost(out.rowPtr, out.colInd, out.values, m, m)

    M1 = run(1)
    P1 = int(po.row_flops(M0, M0).sum())
    P2 = int(po.row_flops(M0, M1).sum())
    assert P2 > P1                                                        # the products grow from iteration 1 to 2
    monkeypatch.setenv("SPGEMM_RMCL_MAXP", str((P1 + P2) // 2))
    M2 = run(2)
    assert h.stats()["nnzC"] >= 0                                         # the last step counted its product: SpGEMM + prune ran
    assert_rmcl_step(M2, M0, M1, what="iteration 2 on the two-step path behind an unpacked Mt")
    M3 = run(3)
    assert_rmcl_step(M3, M0, M2, what="iteration 3")
    monkeypatch.delenv("SPGEMM_RMCL_MAXP")
    dM.deviceDispose()
    h.close()


def test_device_loop_degenerate_inputs_and_argument_errors():
    """hip_gpuRmclIter_device where the fused step gives up (a matrix with no entries: every iteration takes the two-step
    path and returns a packed, empty Mt; a 1x1 graph), and its argument checks (non-square, negative iteration count)."""
    empty = hs.CSR.from_arrays(np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), 5, 5).toGpuCSR()
    out = hs.gpuRmclIter_device(3, empty, empty)
    host = out.toCpuCSR()
    assert host.nnz == 0 and host.rowPtr.tolist() == [0] * 6
    out.deviceDispose()
    one = hs.CSR.from_arrays(np.array([0, 1], np.int32), np.zeros(1, np.int32), np.array([0.5], np.float32), 1, 1).toGpuCSR()
    out = hs.gpuRmclIter_device(4, one, one)
    host = out.toCpuCSR()
    assert host.rowPtr.tolist() == [0, 1] and host.colInd.tolist() == [0] and abs(float(host.values[0]) - 1.0) < 1e-6
    out.deviceDispose()
    with pytest.raises(hs.SpgemmError):
        hs.gpuRmclIter_device(-1, one, one)
    rect = hs.CSR.from_arrays(np.array([0, 1], np.int32), np.zeros(1, np.int32), np.array([0.5], np.float32), 1, 3).toGpuCSR()
    with pytest.raises(hs.SpgemmError):
        hs.gpuRmclIter_device(1, rect, rect)
    for d in (empty, one, rect):
        d.deviceDispose()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_fused_expand_prune_random_rectangular(seed):
    """Random rectangular A (m x k) and B (k x n) with unsorted rows and non-negative values (what R-MCL multiplies;
    with mixed signs a product can be the small difference of large terms and no summation order is within a relative
    bound of another -- the SpGEMM parity rule of DESIGN.md section 2): the fused step against the oracle step."""
    import torch
    from helpers import random_csr
    from sparse_matrix_with_flops_amd.dist import HipEngine, make_matrix
    rng = np.random.default_rng(seed)
    m, k, n = int(rng.integers(50, 400)), int(rng.integers(50, 400)), int(rng.integers(50, 3000))
    A = random_csr(m, k, float(rng.uniform(0.01, 0.2)), 100 + seed, sorted_rows=False, signed=False)
    B = random_csr(k, n, float(rng.uniform(0.01, 0.2)), 200 + seed, sorted_rows=False, signed=False)
    eng = HipEngine(0)
    dA = make_matrix(eng, A.rowPtr, A.colInd, A.values, A.rows, A.cols)
    dB = make_matrix(eng, B.rowPtr, B.colInd, B.values, B.rows, B.cols)
    rp, ci, v = eng.expand_prune(dA, dB)
    torch.cuda.synchronize()
    got = po.CSRHost(rp.cpu().numpy(), ci.cpu().numpy(), v.cpu().numpy(), m, n)
    assert_rmcl_step(got, A, B, what=f"seed {seed}")
