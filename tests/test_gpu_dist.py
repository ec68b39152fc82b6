"""GPU (-m gpu): the N>1 path with the REAL engine.  Two ranks share the one GPU of the test box and talk over gloo
(RCCL refuses two ranks on one device): dist.ShardedSpGEMM / dist.ShardedRMCL run their whole control flow on device
tensors with libspgemm_hip.so doing the local work on every rank -- everything of the multi-GPU step except the RCCL
transport itself, which only an 8-GPU node can exercise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import assert_parity, po, synth_csr

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(m, seed):
    A = synth_csr(m, seed, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    return po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))


def _worker(rank, world, port, kind, m, seed, q, chunks=1, partition="flops"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from sparse_matrix_with_flops_amd.dist import HipEngine, ShardedRMCL, ShardedSpGEMM
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = HipEngine(0, handles=chunks)
        if kind == "spgemm":
            A = synth_csr(m, seed, 2)
            job = ShardedSpGEMM(eng, (A.rowPtr, A.colInd, A.values, A.rows, A.cols), None, chunks=chunks, partition=partition)
            rp, jc, cv = job.step()
            rp, jc, cv = job.step()                   # again: gathered buffers and handles are reused
            torch.cuda.synchronize()
            q.put((rank, rp.cpu().numpy(), jc.cpu().numpy(), cv.cpu().numpy(), job.ends.copy()))
        else:
            Mt = _graph(m, seed)
            host = (Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
            job = ShardedRMCL(eng, host, host)
            states = []
            for _ in range(3):                        # one iteration at a time: every step is checked from the previous state
                job.iterate(1)
                states.append(job.result_host())
            q.put((rank, states, job.ends.copy()))
    finally:
        dist.destroy_process_group()


def _run(kind, m, seed, world=2, chunks=1, partition="flops"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, m, seed, q, chunks, partition)) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    for _ in range(world):                            # a dead worker must fail the test at once, not after a long timeout
        for _try in range(300):
            try:
                outs.append(q.get(timeout=1))
                break
            except Exception:
                if any(p.exitcode not in (None, 0) for p in procs):
                    raise AssertionError(f"a rank died: exit codes {[p.exitcode for p in procs]}")
        else:
            raise AssertionError("timeout waiting for the ranks")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs, key=lambda o: o[0])


@pytest.mark.parametrize("chunks", [1, 3])
def test_sharded_spgemm_two_ranks_hip_engine(chunks):
    """chunks = 3: the sub-block path (one handle per sub-block, exchange of block k overlapping numeric of k+1)."""
    m, seed = 40000, 19
    outs = _run("spgemm", m, seed, chunks=chunks)
    A = synth_csr(m, seed, 2)
    want = po.omp_spmm(A, A)
    flops = po.row_flops(A, A)
    prefix = np.concatenate([[0], np.cumsum(flops)]).astype(np.int64)
    for rank, rp, jc, cv, ends in outs:
        assert np.array_equal(ends, po.equal_partition64(prefix, 2))                # arrayEqualPartition64 on device flops
        assert_parity(po.CSRHost(rp, jc, cv, m, m), want, what=f"rank {rank}")      # every rank holds the whole C


def test_sharded_spgemm_footprint_partition():
    """The reference's other load measure as the cut (static scheduler footprints, static_omp_csr_kernel.cc:28-95):
    flops from hip_csr_row_flops + row lengths of C from a device symbolic pass must give the partition the oracle's
    restatement (pinned to the real reference in tests/test_oracle_vs_ref.py) gives; the product is unchanged."""
    from sparse_matrix_with_flops_amd.dist import equal_partition64
    m, seed = 30000, 21
    outs = _run("spgemm", m, seed, partition="footprint")
    A = synth_csr(m, seed, 2)
    want = po.omp_spmm(A, A)
    cut = equal_partition64(po.footprints(A, A, np.diff(want.rowPtr)), 2)
    for rank, rp, jc, cv, ends in outs:
        assert np.array_equal(ends, cut)
        assert_parity(po.CSRHost(rp, jc, cv, m, m), want, what=f"rank {rank}")


def test_sharded_rmcl_two_ranks_hip_engine():
    """Two ranks, real engine: every rank holds the same replicated Mt after every iteration (bit for bit), and every
    iteration is the oracle's step from the previous state up to counted threshold ties (assert_rmcl_step: identical rows
    at 3e-6, differing rows proven to be ties) -- a defect touching a few rows of the gathered path fails here."""
    from helpers import assert_rmcl_step
    m, seed = 20000, 31
    outs = _run("rmcl", m, seed)
    Mt = _graph(m, seed)
    r0 = outs[0]
    for rank, states, ends in outs:                                                   # all ranks agree bit for bit
        for (rp, ci, v), (rp0, ci0, v0) in zip(states, r0[1]):
            assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0) and np.array_equal(v.view(np.uint32), v0.view(np.uint32))
    cur = Mt
    for k, (rp, ci, v) in enumerate(r0[1]):
        nxt = po.CSRHost(rp, ci, v, m, m)
        ndiff, ties, _ = assert_rmcl_step(nxt, Mt, cur, what=f"two ranks, iteration {k + 1}")
        gl = np.diff(rp)
        rs = np.add.reduceat(v.astype(np.float64), rp[:-1][gl > 0])
        assert np.allclose(rs, 1.0, atol=1e-5)
        cur = nxt
