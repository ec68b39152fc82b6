"""GPU (-m gpu): multi-GPU BEHIND THE C ABI (include/spgemm_hip.h "multi-GPU"; csrc/sharded.hpp) on the one GPU of the test
box: groups of LOGICAL shards that share the device run the whole sharded path -- flops-balanced row cut, per-shard
handles and streams, numeric phase writing into the slice of the gathered arrays, exchange of the row segments -- over
the transports a one-GPU box can run: PEER (device copies), HOST (staged through pinned memory) and RCCL with one rank
(the rank's segment makes a round trip through ncclSend/ncclRecv to itself).  Results against the oracle.  What only an
8-GPU node can show -- the RCCL exchange between different devices -- is the same code path with peers != self."""
import numpy as np
import pytest

from helpers import assert_parity, assert_rmcl_step, po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_parity import to_hs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert hs.device_count() >= 1


def _graph(m, seed):
    A = synth_csr(m, seed, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    return po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))


@pytest.mark.parametrize("shards,transport", [(1, hs.XCHG_PEER), (2, hs.XCHG_PEER), (3, hs.XCHG_HOST), (5, hs.XCHG_PEER),
                                              (1, hs.XCHG_RCCL)])
def test_sharded_spmm_logical_shards(shards, transport):
    m, seed = 40000, 19
    A = synth_csr(m, seed, 2)
    want = po.omp_spmm(A, A)
    g = hs.Group(shards, devices=[0] * shards, transport=transport)
    assert (g.nranks, g.nlocal, g.transport) == (shards, shards, transport)
    job = hs.ShardedSpMM(g, to_hs(A))
    for _ in range(2):                                # twice: buffers and handles are reused
        nnz, P = job.step(gather=True)
    flops = po.row_flops(A, A)
    prefix = np.concatenate([[0], np.cumsum(flops)]).astype(np.int64)
    assert P == int(prefix[-1]) and nnz == want.nnz
    info = job.info()
    assert np.array_equal(info["ends"], po.equal_partition64(prefix, shards))       # arrayEqualPartition64
    for s in range(shards):                           # EVERY shard holds the whole C
        got = job.result(s)
        assert_parity(got, want, what=f"{shards} shards over {hs.XCHG_NAMES[transport]}, shard {s}")
    # not gathered: every shard keeps its own block
    nnz2, _ = job.step(gather=False)
    assert nnz2 == want.nnz
    ends = info["ends"]
    for s in range(shards):
        blk = job.result(s)
        r0, r1 = ends[s], ends[s + 1]
        assert blk.rows == r1 - r0
        sub = po.CSRHost(want.rowPtr[r0:r1 + 1] - want.rowPtr[r0], want.colInd[want.rowPtr[r0]:want.rowPtr[r1]],
                         want.values[want.rowPtr[r0]:want.rowPtr[r1]], r1 - r0, m)
        assert_parity(blk, sub, what=f"block of shard {s}")
    job.close()
    g.close()


def test_sharded_spmm_rectangular_and_empty_blocks():
    """A != B (rectangular) and more shards than rows with products: empty blocks and empty segments travel too."""
    from helpers import random_csr
    A = random_csr(37, 200, 0.05, 3, sorted_rows=False, signed=False)
    B = random_csr(200, 5000, 0.01, 4, sorted_rows=False, signed=False)
    want = po.omp_spmm(A, B)
    for shards, tr in ((4, hs.XCHG_PEER), (7, hs.XCHG_HOST)):
        g = hs.Group(shards, devices=[0] * shards, transport=tr)
        job = hs.ShardedSpMM(g, to_hs(A), to_hs(B))
        nnz, _ = job.step()
        assert nnz == want.nnz
        for s in (0, shards - 1):
            assert_parity(job.result(s), want, what=f"rectangular, {shards} shards, shard {s}")
        job.close()
        g.close()


@pytest.mark.parametrize("shards,transport", [(2, hs.XCHG_PEER), (3, hs.XCHG_HOST), (1, hs.XCHG_RCCL)])
def test_sharded_rmcl_steps_match_the_oracle(shards, transport):
    """hip_gpuRmclIter_sharded one iteration at a time: every step from the previous state is the oracle's step up to
    counted threshold ties (assert_rmcl_step) -- prune before gather, offsets, exchange."""
    m, seed = 20000, 31
    Mt = _graph(m, seed)
    g = hs.Group(shards, devices=[0] * shards, transport=transport)
    Mg = to_hs(Mt)
    cur = Mt
    for k in range(3):
        nxt = hs.gpuRmclIter_sharded(g, 1, Mg, to_hs(cur))
        nxt_h = po.CSRHost(nxt.rowPtr, nxt.colInd, nxt.values, m, m)
        ndiff, ties, _ = assert_rmcl_step(nxt_h, Mt, cur, what=f"{shards} shards over {hs.XCHG_NAMES[transport]}, iteration {k + 1}")
        gl = np.diff(nxt.rowPtr)
        rs = np.add.reduceat(nxt.values.astype(np.float64), nxt.rowPtr[:-1][gl > 0])
        assert np.allclose(rs, 1.0, atol=1e-5)
        cur = nxt_h
    # several iterations inside one call give the same matrix as the single-device entry point (up to ties: the fixture
    # graph of the reference has none)
    g.close()


def test_gpuRmclIter_dispatches_to_the_sharded_loop(monkeypatch):
    """hip_gpuRmclIter itself goes multi-shard WHEN ASKED (SPGEMM_RMCL_DEVICES=N|all: that many devices;
    SPGEMM_RMCL_SHARDS: logical shards on them) -- opt-in since round 4: unset, it computes on one device whatever is
    visible.  Same result as the single-device loop on a tie-free input; the group is kept between calls."""
    import os
    from helpers import DATA
    Mt = po.load(os.path.join(DATA, "own_graph.snap"), isTrans=True, mode=1)
    monkeypatch.delenv("SPGEMM_RMCL_SHARDS", raising=False)
    monkeypatch.delenv("SPGEMM_RMCL_DEVICES", raising=False)
    one = hs.gpuRmclIter(3, to_hs(Mt), to_hs(Mt))
    assert hs.lib().spgemm_hip_rmcl_devices_used() == 1            # nothing asked for: one device
    monkeypatch.setenv("SPGEMM_RMCL_DEVICES", "all")
    hs.gpuRmclIter(1, to_hs(Mt), to_hs(Mt))
    assert hs.lib().spgemm_hip_rmcl_devices_used() == hs.device_count()
    monkeypatch.setenv("SPGEMM_RMCL_DEVICES", "1")
    monkeypatch.setenv("SPGEMM_RMCL_SHARDS", "3")
    three = hs.gpuRmclIter(3, to_hs(Mt), to_hs(Mt))
    again = hs.gpuRmclIter(3, to_hs(Mt), to_hs(Mt))               # the cached group
    assert np.array_equal(again.rowPtr, three.rowPtr) and np.array_equal(again.colInd, three.colInd)
    a, b = po.CSRHost(one.rowPtr, one.colInd, one.values, Mt.rows, Mt.cols), po.CSRHost(three.rowPtr, three.colInd, three.values, Mt.rows, Mt.cols)
    assert_parity(b, a, what="3 logical shards vs one device")


def test_group_argument_errors():
    with pytest.raises(hs.SpgemmError):
        hs.Group(0)
    with pytest.raises(hs.SpgemmError):
        hs.Group(2, devices=[0, 99])
    with pytest.raises(hs.SpgemmError):
        hs.Group(2, devices=[0, 0], transport=hs.XCHG_RCCL)       # RCCL needs one device per shard
    ident = hs.unique_id()
    assert len(ident) == hs.UNIQUE_ID_BYTES
    g = hs.Group.of_rank(1, 0, 0, ident)                          # the multi-process constructor with a world of one
    assert (g.nranks, g.nlocal, g.transport) == (1, 1, hs.XCHG_RCCL)
    A = synth_csr(3000, 5, 2)
    job = hs.ShardedSpMM(g, to_hs(A))
    nnz, _ = job.step()
    assert_parity(job.result(0), po.omp_spmm(A, A), what="one rank through create_rank")
    job.close()
    g.close()


def test_cpp_driver_runs_the_sharded_loop():
    """The C++ mirror's nrmcl driver (reference flags, Same/Diffs against the CPU checker) with the R-MCL loop cut into
    logical shards behind the one call it makes -- gpuRmclIter (nlibs/qrmcl.cc:149-152)."""
    import os
    import subprocess
    from helpers import DATA, ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    exe = os.path.join(ROOT, "tests", "cpp", "nrmcl.x")
    env = dict(os.environ, SPGEMM_RMCL_SHARDS="2")
    out = subprocess.run([exe, "-i", os.path.join(DATA, "own_graph.snap"), "-m", "3", "-r", "GPU"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Same" in out.stdout and "Diffs" not in out.stdout
    assert "time pass readSNAPFile" in out.stdout and "time pass gpuRmclIter" in out.stdout


def test_dist_module_calls_the_library_entry_points():
    """dist.library_group / LibraryShardedSpGEMM / library_rmcl -- what bench.py --gpus N runs on every rank -- with a world
    of one: the group comes from spgemm_hip_group_create_rank, the segment makes its round trip through RCCL."""
    import torch  # noqa: F401  (the torch runtime first: tests/conftest.py)
    from sparse_matrix_with_flops_amd.dist import LibraryShardedSpGEMM, library_group, library_rmcl
    A = synth_csr(30000, 41, 2)
    g = library_group(0)
    assert (g.nranks, g.nlocal, g.transport) == (1, 1, hs.XCHG_RCCL)
    job = LibraryShardedSpGEMM(g, (A.rowPtr, A.colInd, A.values, A.rows, A.cols))
    want = po.omp_spmm(A, A)
    assert job.step(gather=True) == want.nnz and job.total_flops == int(po.row_flops(A, A).sum())
    assert_parity(job.result_host(), want, what="LibraryShardedSpGEMM, world 1")
    st = job.handle.stats()
    assert st["nnzC"] == want.nnz and job.info()["ends"] == [0, A.rows]
    Mt = _graph(8000, 13)
    host = (Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
    nxt = library_rmcl(g, 1, host, host)
    assert_rmcl_step(po.CSRHost(nxt.rowPtr, nxt.colInd, nxt.values, Mt.rows, Mt.cols), Mt, Mt, what="library_rmcl, world 1")
    g.close()


def test_a_stale_hip_error_of_another_library_does_not_fail_the_next_call():
    """RCCL probes devices when a communicator is made and leaves "invalid device ordinal" in the thread's last-error slot;
    the library checks that slot after its launch sequences.  Seen once as a spurious failure of the next hip_gpuSpMM
    (test order dependent): every launch sequence now empties the slot first.  Make a communicator, then multiply."""
    g = hs.Group(1, devices=[0], transport=hs.XCHG_RCCL)
    A = synth_csr(5000, 3, 2)
    dA = to_hs(A).toGpuCSR()
    h = hs.Handle(0)
    dC = hs.gpuSpMMWrapper(dA, dA, h)                  # failed with "hipGetLastError() failed: invalid device ordinal"
    got = dC.toCpuCSR()
    dC.deviceDispose()
    dA.deviceDispose()
    assert_parity(got, po.omp_spmm(A, A), what="SpGEMM right after RCCL initialisation")
    h.close()
    g.close()


_LOOPBACK_SCRIPT = r"""
import os, sys, threading
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from helpers import assert_parity, assert_rmcl_step, po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_sharded_abi import _graph, to_hs_plain
W = {world}
A = synth_csr(30000, 23, 2)
want = po.omp_spmm(A, A)
Mt = _graph(12000, 31)
ident = hs.unique_id()
res, errs = [None] * W, []
def rank(r):
    try:
        g = hs.Group.of_rank(W, r, 0, ident)                 # blocks until every rank has joined, like ncclCommInitRank
        assert (g.nranks, g.nlocal, g.transport) == (W, 1, hs.XCHG_RCCL)
        job = hs.ShardedSpMM(g, to_hs_plain(A))
        for _ in range(2):
            nnz, P = job.step(gather=True)
        full = job.result(0)
        nnz_blk, _ = job.step(gather=False)
        blk = job.result(0)
        info = job.info()
        job.close()
        nxt = hs.gpuRmclIter_sharded(g, 2, to_hs_plain(Mt), to_hs_plain(Mt))
        g.close()
        res[r] = (nnz, P, full, nnz_blk, blk, info, nxt)
    except BaseException as e:                               # noqa: BLE001
        errs.append((r, repr(e)))
ts = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
[t.start() for t in ts]
[t.join() for t in ts]
assert not errs, errs
flops = po.row_flops(A, A)
prefix = np.concatenate([[0], np.cumsum(flops)]).astype(np.int64)
ends = po.equal_partition64(prefix, W)
one = None
for r in range(W):
    nnz, P, full, nnz_blk, blk, info, nxt = res[r]
    assert nnz == want.nnz and P == int(prefix[-1]) and np.array_equal(info["ends"], ends)
    assert_parity(full, want, what="rank %d of %d: gathered C" % (r, W))      # EVERY rank holds the whole product
    r0, r1 = ends[r], ends[r + 1]
    assert blk.rows == r1 - r0
    sub = po.CSRHost(want.rowPtr[r0:r1 + 1] - want.rowPtr[r0], want.colInd[want.rowPtr[r0]:want.rowPtr[r1]],
                     want.values[want.rowPtr[r0]:want.rowPtr[r1]], r1 - r0, A.cols)
    assert_parity(blk, sub, what="rank %d: own block" % r)
    got = po.CSRHost(nxt.rowPtr, nxt.colInd, nxt.values, Mt.rows, Mt.cols)
    if one is None:
        one = got
        first = hs.gpuRmclIter_sharded(hs.Group(1, devices=[0], transport=hs.XCHG_PEER), 1, to_hs_plain(Mt), to_hs_plain(Mt))
        assert_rmcl_step(got, Mt, po.CSRHost(first.rowPtr, first.colInd, first.values, Mt.rows, Mt.cols), what="2 iterations over %d ranks" % W)
    else:                                                    # every rank returns the SAME matrix (it was gathered)
        assert np.array_equal(got.rowPtr, one.rowPtr) and np.array_equal(got.colInd, one.colInd) and np.array_equal(got.values, one.values)
maps = open("/proc/self/maps").read()                      # the exchange really went through the stand-in
assert "libloopback_rccl.so" in maps and "librccl.so" not in maps
print("loopback ok", W)
"""


def to_hs_plain(M):
    return hs.CSR.from_arrays(M.rowPtr, M.colInd, M.values, M.rows, M.cols)


@pytest.mark.parametrize("world", [2, 3])
def test_rank_groups_exchange_over_a_loopback_rccl(world):
    """The multi-PROCESS form of the group (spgemm_hip_group_create_rank: one shard per rank, RCCL the only transport) with
    peers != self -- what bench.py --gpus N and an 8-GPU node run -- cannot execute on a one-GPU box with the real RCCL
    (two ranks on one device are refused).  tests/cpp/loopback_rccl.cc stands in for the nine RCCL entry points the library
    loads (SPGEMM_RCCL_LIB): ranks are THREADS sharing the GPU, sends and receives are matched per pair in posting order
    as NCCL does, a size mismatch is an error and an unmatched receive blocks.  Checked: the communicator set-up from a
    distributed id, ncclAllGather of the segment sizes, the grouped ncclSend/ncclRecv of the three arrays per peer with
    their counts and offsets, the R-MCL loop's gather of pruned blocks -- every rank ends with the oracle's product.
    Own process: the library loads its RCCL once."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "libloopback_rccl.so"])
    env = dict(os.environ, SPGEMM_RCCL_LIB=os.path.join(ROOT, "tests", "cpp", "libloopback_rccl.so"))
    out = subprocess.run([sys.executable, "-c", _LOOPBACK_SCRIPT.format(root=ROOT, world=world)], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 0 and f"loopback ok {world}" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


_FAIL_SCRIPT = r"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from helpers import po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_sharded_abi import _graph, to_hs_plain
W, BAD = {world}, {bad}
A = synth_csr(20000, 29, 2)
Mt = _graph(9000, 37)
ident = hs.unique_id()
out, errs = [None] * W, []
def rank(r):
    try:
        g = hs.Group.of_rank(W, r, 0, ident)
        job = hs.ShardedSpMM(g, to_hs_plain(A))
        job.step(gather=True)                                  # a good step first
        if r == BAD:
            job.handle(0).fail_next(1)                         # this rank's next symbolic phase fails before it queues anything
        t0 = time.time()
        try:
            job.step(gather=True)
            first = "no error"
        except hs.SpgemmError as e:
            first = str(e)
        dt = time.time() - t0
        nnz, _ = job.step(gather=True)                         # and the job is usable again on every rank
        rm = hs.ShardedRmcl(g, to_hs_plain(Mt), to_hs_plain(Mt))
        rm.run(1)
        if r == BAD:
            # (a handle of the group: the R-MCL job computes with the same one)
            job.handle(0).fail_next(1)
        try:
            rm.run(2)
            second = "no error"
        except hs.SpgemmError as e:
            second = str(e)
        n2 = rm.run(2)
        rm.close(); job.close(); g.close()                     # the group stays destroyable
        out[r] = (first, dt, nnz, second, n2)
    except BaseException as e:                                 # noqa: BLE001
        errs.append((r, repr(e)))
ts = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
[t.start() for t in ts]
[t.join(120) for t in ts]
assert not any(t.is_alive() for t in ts), "a rank is still waiting in a collective"
assert not errs, errs
want = po.omp_spmm(A, A).nnz
for r in range(W):
    first, dt, nnz, second, n2 = out[r]
    assert first != "no error" and dt < 30.0, (r, first, dt)
    assert ("forced failure" in first) == (r == BAD) and (r == BAD or "rank %d failed" % BAD in first), (r, first)
    assert nnz == want
    assert second != "no error" and (("forced failure" in second) == (r == BAD)), (r, second)
    assert n2 == out[0][4]
print("failure propagated", W)
"""


@pytest.mark.parametrize("world,bad", [(2, 1), (3, 0)])
def test_a_failing_rank_takes_every_rank_out_of_the_step(world, bad):
    """Round-3 review: a rank whose symbolic phase failed returned before the size exchange and left the other ranks waiting
    in ncclAllGather forever.  Now the failing rank enters the exchange with a sentinel: EVERY rank returns an error (the
    failing one its own, the others "rank r failed"), within a timeout, the job and the group stay usable and destroyable --
    for the sharded SpGEMM step and for the sharded R-MCL loop.  Ranks = threads over the loopback stand-in for RCCL."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "libloopback_rccl.so"])
    env = dict(os.environ, SPGEMM_RCCL_LIB=os.path.join(ROOT, "tests", "cpp", "libloopback_rccl.so"))
    out = subprocess.run([sys.executable, "-c", _FAIL_SCRIPT.format(root=ROOT, world=world, bad=bad)], capture_output=True,
                         text=True, timeout=400, env=env)
    assert out.returncode == 0 and f"failure propagated {world}" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("shards,transport", [(1, hs.XCHG_PEER), (3, hs.XCHG_PEER), (2, hs.XCHG_HOST), (1, hs.XCHG_RCCL)])
def test_sharded_rmcl_job_keeps_its_operands_resident(shards, transport):
    """hip_sharded_rmcl_create / run / result (round 4): the loop on device arrays only -- what bench.py times at N > 1.
    Every run starts from the initial Mt (same result twice), every iteration is checked against the oracle's step from
    the previous iteration's result, nnz per iteration is reported, every shard holds the whole result, and a failure inside
    a run leaves the job usable."""
    Mt = _graph(15000, 41)
    g = hs.Group(shards, devices=[0] * shards, transport=transport)
    job = hs.ShardedRmcl(g, to_hs_plain(Mt), to_hs_plain(Mt))
    assert job.run(0) == Mt.nnz
    r0 = job.result(0)
    assert np.array_equal(r0.rowPtr, Mt.rowPtr) and np.array_equal(r0.colInd, Mt.colInd)
    prev = Mt
    for iters in (1, 2, 3):
        n = job.run(iters)
        cur = job.result(0)
        assert n == cur.nnz and job.iter_nnz()[-1] == n and len(job.iter_nnz()) == iters
        got = po.CSRHost(cur.rowPtr, cur.colInd, cur.values, Mt.rows, Mt.cols)
        assert_rmcl_step(got, Mt, prev, what=f"iteration {iters} of the resident loop, {shards} shards")
        prev = got
    job.run(2)                                         # one step of the SAME trajectory (restart=False): exact bookkeeping
    two = job.result(0)
    n3 = job.run(1, restart=False)
    three = job.result(0)
    assert n3 == three.nnz and job.iter_nnz()[-1] == n3 and len(job.iter_nnz()) == 3
    assert_rmcl_step(po.CSRHost(three.rowPtr, three.colInd, three.values, Mt.rows, Mt.cols), Mt,
                     po.CSRHost(two.rowPtr, two.colInd, two.values, Mt.rows, Mt.cols), what="continued run")
    again = job.run(3)
    assert again == prev.nnz
    want_c = prev.canonical()
    for s_ in range(shards):                           # (entries inside a row come in table order: compare the canonical forms)
        rs = job.result(s_)
        got_c = po.CSRHost(rs.rowPtr, rs.colInd, rs.values, Mt.rows, Mt.cols).canonical()
        assert np.array_equal(got_c.rowPtr, want_c.rowPtr) and np.array_equal(got_c.colInd, want_c.colInd)
        assert np.allclose(got_c.values, want_c.values, rtol=3e-6, atol=0.0)
    flops = po.row_flops(Mt, Mt)
    prefix = np.concatenate([[0], np.cumsum(flops)]).astype(np.int64)
    assert np.array_equal(job.ends(), po.equal_partition64(prefix, shards))
    g.close()                                          # closes the job first (Group keeps track of what was made on it)
    with pytest.raises(Exception):
        job.run(1)
