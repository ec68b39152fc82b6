"""CPU, world_size 2 (and 3) over gloo: the multi-GPU control flow of sparse_matrix_with_flops_amd/dist.py —
flops-balanced row partition, per-rank symbolic/numeric into a slice of the gathered buffers, allgatherv by
send/recv pairs — with the local compute swapped for the CPU oracle (tests may use the oracle; the product's
engine is HipEngine and has no CPU fallback)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import assert_parity, po, synth_csr
from sparse_matrix_with_flops_amd.dist import ShardedRMCL, ShardedSpGEMM, equal_partition64


class OracleEngine:
    """Same interface as dist.HipEngine, on CPU tensors, computing with oracle/ (test double)."""

    def __init__(self, handles=1):
        self.handles = [None] * handles           # one pending symbolic phase per slot, like HipEngine
        self._C = [None] * handles

    def tensor(self, arr, dtype):
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype)

    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype)

    def sync(self):
        pass

    @staticmethod
    def _host(M):
        return po.CSRHost(M["rowPtr"].numpy(), M["colInd"].numpy(), M["values"].numpy(), M["rows"], M["cols"])

    def row_flops(self, A, B):
        return po.row_flops(self._host(A), self._host(B))

    def symbolic(self, A, B, slot=0):
        self._C[slot] = po.omp_spmm(self._host(A), self._host(B))
        return torch.from_numpy(self._C[slot].rowPtr.copy()), self._C[slot].nnz

    def numeric(self, A, B, IC, JC_out, C_out, slot=0):
        JC_out.copy_(torch.from_numpy(self._C[slot].colInd))
        C_out.copy_(torch.from_numpy(self._C[slot].values))

    def expand_prune(self, A, B):
        R = po.rmcl_iters(self._host(A), self._host(B), 1)               # prune(A*B), rows of A
        return (torch.from_numpy(R.rowPtr.copy()), torch.from_numpy(R.colInd.copy()), torch.from_numpy(R.values.copy()))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, seed, q, chunks=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = synth_csr(m, seed, 2)
        job = ShardedSpGEMM(OracleEngine(chunks), (A.rowPtr, A.colInd, A.values, A.rows, A.cols), None, chunks=chunks)
        rp, jc, cv = job.step()
        rp, jc, cv = job.step()                      # second step: the gathered buffers are reused
        assert job.chunks == chunks
        q.put((rank, rp.numpy().copy(), jc.numpy().copy(), cv.numpy().copy(), job.ends.copy(), job.local_flops))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 1), (2, 3), (3, 2)])
def test_sharded_spgemm_gloo(world, chunks):
    """chunks > 1: every rank's block is cut into sub-blocks whose send/recv pairs overlap the next sub-block's numeric
    phase; the gathered C must not depend on the cut."""
    m, seed = 3000, 19
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, seed, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A = synth_csr(m, seed, 2)
    want = po.sequential_spmm(A, A)
    flops = po.row_flops(A, A)
    for rank, rp, jc, cv, ends, lf in outs:
        got = po.CSRHost(rp, jc, cv, m, m)
        assert_parity(got, want, what=f"rank {rank} of {world}")         # every rank holds the whole C
        assert lf == int(flops[ends[rank]:ends[rank + 1]].sum())
    # balanced: no rank has more than ~1/world + one heavy row of the work
    shares = [o[5] for o in sorted(outs)]
    assert sum(shares) == int(flops.sum()) and max(shares) <= flops.sum() / world + flops.max()


def _rmcl_graph(m, seed):
    A = synth_csr(m, seed, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    return po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))     # transpose + self loops + 1/deg


def _rmcl_worker(rank, world, port, m, seed, iters, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Mt = _rmcl_graph(m, seed)
        host = (Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
        job = ShardedRMCL(OracleEngine(), host, host)
        job.iterate(iters)
        rp, ci, v = job.result_host()
        q.put((rank, rp.copy(), ci.copy(), v.copy(), job.ends.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_rmcl_gloo(world):
    """R-MCL with Mgt row-sharded and the pruned blocks gathered every iteration == the sequential loop, bit for bit
    (the local engine here is the sequential oracle, so even the float bits agree)."""
    m, seed, iters = 1500, 23, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rmcl_worker, args=(r, world, port, m, seed, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Mt = _rmcl_graph(m, seed)
    want = po.rmcl_iters(Mt, Mt, iters)
    for rank, rp, ci, v, ends in outs:
        assert ends[0] == 0 and ends[-1] == m and np.all(np.diff(ends) > 0)
        assert np.array_equal(rp, want.rowPtr) and np.array_equal(ci, want.colInd), f"rank {rank}"
        assert np.array_equal(v.view(np.uint32), want.values.view(np.uint32)), f"rank {rank}"


def test_partition_matches_reference_rule():
    rng = np.random.default_rng(3)
    for n, parts in [(1, 1), (5, 2), (100, 8), (1000, 7), (17, 32)]:
        f = rng.integers(0, 50, size=n)
        prefix = np.concatenate([[0], np.cumsum(f)]).astype(np.int64)
        assert np.array_equal(equal_partition64(prefix, parts), po.equal_partition64(prefix, parts)), (n, parts)
    prefix = np.array([0, 0, 100, 100, 100], dtype=np.int64)
    for parts in (2, 4):
        assert np.array_equal(equal_partition64(prefix, parts), po.equal_partition64(prefix, parts))
