"""Row-sharded multi-GPU SpGEMM: A split into flops-balanced contiguous row blocks, B replicated,
C's row segments exchanged with an allgatherv over RCCL/xGMI (torch.distributed, one process per GPU).

The reference has no multi-device code at all; what it does have is the same idea across
CPU threads: rows are cut into contiguous ranges of equal *flops* with arrayEqualPartition64
(nlibs/tools/util.cc:123-135, used by flops_omp_CSR_SpMM, nlibs/flops_csr_kernel.cc:59-63).  That partition
rule is reused here across GPUs.  C row i depends only on A row i and all of B (Gustavson), so there is
exactly one exchange step, at the end.

xGMI is a full mesh of point-to-point links: the gather is issued as grouped send/recv pairs
(dist.batch_isend_irecv -> ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd), so every peer's segment
travels its own link instead of circulating a ring.

PyTorch is used for device memory and torch.distributed only; the SpGEMM itself is libspgemm_hip.so.
"""
import numpy as np

try:  # torch is plumbing here; the single-GPU C-ABI path works without it
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

from . import hipspgemm as hs


def equal_partition64(prefix, parts):
    """arrayEqualPartition64 (nlibs/tools/util.cc:123-135): `prefix` is the exclusive scan of per-row
    flops with prefix[n] = total.  Returns ends[parts+1]; part p owns rows [ends[p], ends[p+1])."""
    prefix = np.asarray(prefix, dtype=np.int64)
    n = len(prefix) - 1
    total = int(prefix[n])
    chunk = (total + parts - 1) // parts
    ends = np.zeros(parts + 1, dtype=np.int64)
    now = 0
    for i in range(parts - 1):
        target = min((i + 1) * chunk, total)
        upper = now + int(np.searchsorted(prefix[now:n + 1], target, side="right"))   # std::upper_bound
        e = max(upper - 1, now + 1)
        e = min(e, n)
        ends[i + 1] = e
        now = e
    ends[parts] = n
    return ends


def _ptr(t):
    return t.data_ptr() if t.numel() else 0


class HipEngine:
    """Local compute on one GPU through the C ABI (no CPU fallback)."""

    def __init__(self, device_index):
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.handle = hs.Handle(device_index)

    def tensor(self, arr, dtype):
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype).to(self.device)

    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def sync(self):
        torch.cuda.current_stream().synchronize()

    def row_flops(self, A, B):
        """A, B: dicts of device tensors (rowPtr, colInd, values, rows, cols).  -> np.int64[rows]"""
        out = self.empty(A["rows"], torch.int32)
        self.sync()
        hs.row_flops_raw(self.handle, _ptr(A["rowPtr"]), _ptr(A["colInd"]), _ptr(B["rowPtr"]), A["rows"], _ptr(out))
        return out.cpu().numpy().astype(np.int64)

    def symbolic(self, A, B):
        IC = self.empty(A["rows"] + 1, torch.int32)
        self.sync()
        nnz = hs.spgemm_symbolic_raw(self.handle, _ptr(A["rowPtr"]), _ptr(A["colInd"]), A["nnz"], _ptr(B["rowPtr"]),
                                     _ptr(B["colInd"]), B["nnz"], A["rows"], A["cols"], B["cols"], _ptr(IC))
        return IC, nnz

    def numeric(self, A, B, IC, JC_out, C_out):
        self.sync()
        hs.spgemm_numeric_raw(self.handle, _ptr(A["rowPtr"]), _ptr(A["colInd"]), _ptr(A["values"]), A["nnz"],
                              _ptr(B["rowPtr"]), _ptr(B["colInd"]), _ptr(B["values"]), B["nnz"], A["rows"], A["cols"],
                              B["cols"], _ptr(IC), _ptr(JC_out), _ptr(C_out))

    def stats(self):
        return self.handle.stats()


def make_matrix(engine, rowPtr, colInd, values, rows, cols):
    return {"rowPtr": engine.tensor(rowPtr, torch.int32), "colInd": engine.tensor(colInd, torch.int32),
            "values": engine.tensor(values, torch.float32), "rows": int(rows), "cols": int(cols),
            "nnz": int(rowPtr[-1]) if len(rowPtr) else 0}


class ShardedSpGEMM:
    """C = A * B with A row-sharded over the process group.  Every rank ends up with the whole C."""

    def __init__(self, engine, A_host, B_host=None, group=None):
        """A_host/B_host: (rowPtr, colInd, values, rows, cols) numpy tuples, identical on every rank.
        B_host=None means C = A*A (B is the replicated full A)."""
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if (dist and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if (dist and dist.is_initialized()) else 0
        rpA, ciA, vA, mA, kA = A_host
        self.m, self.k = int(mA), int(kA)
        self.B = make_matrix(engine, *(B_host if B_host is not None else A_host))
        self.n = self.B["cols"]
        # flops-balanced contiguous partition, computed identically on every rank
        fullA = self.B if B_host is None else make_matrix(engine, *A_host)
        flops = engine.row_flops(fullA, self.B)
        self.total_flops = int(flops.sum())
        prefix = np.zeros(self.m + 1, dtype=np.int64)
        np.cumsum(flops, out=prefix[1:])
        self.ends = equal_partition64(prefix, self.world)
        r0, r1 = int(self.ends[self.rank]), int(self.ends[self.rank + 1])
        self.r0, self.r1 = r0, r1
        self.local_flops = int(prefix[r1] - prefix[r0])
        rpA = np.asarray(rpA)
        lo, hi = int(rpA[r0]), int(rpA[r1])
        self.A_local = make_matrix(engine, (rpA[r0:r1 + 1] - lo).astype(np.int32), np.asarray(ciA)[lo:hi],
                                   np.asarray(vA)[lo:hi], r1 - r0, self.k)
        del fullA

    # ---- one hot-path pass --------------------------------------------------------------------
    def step(self, gather=True):
        """gather=True: every rank returns the whole C (rowPtr, colInd, values).  gather=False: C stays row-sharded
        like A — returns this rank's (local rowPtr, colInd, values) and no collective runs."""
        eng, G, me = self.engine, self.world, self.rank
        IC_loc, nnz_loc = eng.symbolic(self.A_local, self.B)
        if G == 1 or not gather:
            JC = eng.empty(max(nnz_loc, 1), torch.int32)
            Cv = eng.empty(max(nnz_loc, 1), torch.float32)
            eng.numeric(self.A_local, self.B, IC_loc, JC, Cv)
            return IC_loc, JC[:nnz_loc], Cv[:nnz_loc]
        # (1) segment sizes
        mine = torch.tensor([nnz_loc], dtype=torch.int64, device=IC_loc.device)
        sizes = [torch.zeros(1, dtype=torch.int64, device=IC_loc.device) for _ in range(G)]
        dist.all_gather(sizes, mine, group=self.group)
        counts = [int(s.item()) for s in sizes]
        offs = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(counts, out=offs[1:])
        total = int(offs[G])
        if total > 0x7fffffff:
            raise hs.SpgemmError(f"nnz(C)={total} does not fit the int32 CSR of the boundary")
        # (2) numeric straight into this rank's slice of the gathered arrays
        rowPtr = eng.empty(self.m + 1, torch.int32)
        JC = eng.empty(max(total, 1), torch.int32)
        Cv = eng.empty(max(total, 1), torch.float32)
        o0, o1 = int(offs[me]), int(offs[me + 1])
        eng.numeric(self.A_local, self.B, IC_loc, JC[o0:o1] if o1 > o0 else JC[0:0], Cv[o0:o1] if o1 > o0 else Cv[0:0])
        rowPtr[self.r0:self.r1] = IC_loc[:-1] + o0
        if me == G - 1:
            rowPtr[self.m] = total
        # (3) allgatherv of the three arrays: pairwise send/recv, one link per peer
        ops = []
        for r in range(G):
            if r == me:
                continue
            a0, a1 = int(offs[r]), int(offs[r + 1])
            rr0, rr1 = int(self.ends[r]), int(self.ends[r + 1]) + (1 if r == G - 1 else 0)
            my_rows_hi = self.r1 + (1 if me == G - 1 else 0)
            if my_rows_hi > self.r0:
                ops.append(dist.P2POp(dist.isend, rowPtr[self.r0:my_rows_hi], r, self.group))
            if rr1 > rr0:
                ops.append(dist.P2POp(dist.irecv, rowPtr[rr0:rr1], r, self.group))
            if o1 > o0:
                ops.append(dist.P2POp(dist.isend, JC[o0:o1], r, self.group))
                ops.append(dist.P2POp(dist.isend, Cv[o0:o1], r, self.group))
            if a1 > a0:
                ops.append(dist.P2POp(dist.irecv, JC[a0:a1], r, self.group))
                ops.append(dist.P2POp(dist.irecv, Cv[a0:a1], r, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return rowPtr, JC[:total], Cv[:total]
