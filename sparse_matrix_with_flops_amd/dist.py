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

    def expand_prune(self, A, B):
        """One R-MCL step on this rank's rows: C = A*B (hip_gpuSpMM), then inflate/prune/normalise + compaction
        (hip_rmcl_prune), all on the device.  Returns torch tensors (rowPtr[rows+1], colInd, values)."""
        self.sync()
        ic, jc, cv, _ = hs.gpu_spmm_raw(self.handle, _ptr(A["rowPtr"]), _ptr(A["colInd"]), _ptr(A["values"]), A["nnz"],
                                        _ptr(B["rowPtr"]), _ptr(B["colInd"]), _ptr(B["values"]), B["nnz"],
                                        A["rows"], A["cols"], B["cols"])
        try:
            pi, pj, pv, nn = hs.rmcl_prune_raw(self.handle, A["rows"], ic, jc, cv)
        finally:
            for p in (ic, jc, cv):
                hs.dev_free(p)
        try:
            rp = self.empty(A["rows"] + 1, torch.int32)
            ci = self.empty(max(nn, 1), torch.int32)
            v = self.empty(max(nn, 1), torch.float32)
            hs.d2d(rp.data_ptr(), pi, 4 * (A["rows"] + 1))
            hs.d2d(ci.data_ptr(), pj, 4 * nn)
            hs.d2d(v.data_ptr(), pv, 4 * nn)
        finally:
            for p in (pi, pj, pv):
                hs.dev_free(p)
        return rp, ci[:nn], v[:nn]

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
        offs = _segment_offsets(self.group, G, nnz_loc, IC_loc.device)
        total = int(offs[G])
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
        _allgatherv_csr(self.group, me, G, self.ends, self.m, offs, rowPtr, JC, Cv)
        return rowPtr, JC[:total], Cv[:total]


def _allgatherv_csr(group, me, G, ends, m, offs, rowPtr, JC, Cv):
    """Every rank has filled its own rows of rowPtr (global offsets; the last rank also rowPtr[m]) and its own
    segment [offs[me], offs[me+1]) of JC/Cv; exchange so that all ranks hold everything.  Grouped isend/irecv pairs:
    each peer's segment travels its own xGMI link."""
    r0, r1 = int(ends[me]), int(ends[me + 1])
    o0, o1 = int(offs[me]), int(offs[me + 1])
    my_rows_hi = r1 + (1 if me == G - 1 else 0)
    ops = []
    for r in range(G):
        if r == me:
            continue
        a0, a1 = int(offs[r]), int(offs[r + 1])
        rr0, rr1 = int(ends[r]), int(ends[r + 1]) + (1 if r == G - 1 else 0)
        if my_rows_hi > r0:
            ops.append(dist.P2POp(dist.isend, rowPtr[r0:my_rows_hi], r, group))
        if rr1 > rr0:
            ops.append(dist.P2POp(dist.irecv, rowPtr[rr0:rr1], r, group))
        if o1 > o0:
            ops.append(dist.P2POp(dist.isend, JC[o0:o1], r, group))
            ops.append(dist.P2POp(dist.isend, Cv[o0:o1], r, group))
        if a1 > a0:
            ops.append(dist.P2POp(dist.irecv, JC[a0:a1], r, group))
            ops.append(dist.P2POp(dist.irecv, Cv[a0:a1], r, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def _segment_offsets(group, G, nnz_loc, device):
    """all-gather of the per-rank segment sizes -> offs[G+1] (int64 numpy)"""
    mine = torch.tensor([nnz_loc], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(G)]
    dist.all_gather(sizes, mine, group=group)
    offs = np.zeros(G + 1, dtype=np.int64)
    np.cumsum([int(x.item()) for x in sizes], out=offs[1:])
    if int(offs[G]) > 0x7fffffff:
        raise hs.SpgemmError(f"nnz={int(offs[G])} does not fit the int32 CSR of the boundary")
    return offs


class ShardedRMCL:
    """R-MCL over the process group (BASELINE configs[4]): Mt <- prune(Mgt * Mt), maxIter times
    (nlibs/qrmcl.cc:86-124; the reference's GPU loop is single-device, nlibs/gpus/gpu_csr_kernel.cu:281-311).
    Mgt's row blocks (cut once, by the flops of the first expansion) stay resident per GPU; Mt is replicated.
    Inflate/prune/normalise are row-local, so they run BEFORE the gather: what crosses xGMI is the pruned block,
    several times smaller than the raw product."""

    def __init__(self, engine, Mgt_host, Mt_host, group=None):
        """Mgt_host/Mt_host: (rowPtr, colInd, values, rows, cols) numpy tuples, identical on every rank."""
        self.engine = engine
        self.group = group
        on = bool(dist and dist.is_initialized())
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        rpG, ciG, vG, mG, kG = Mgt_host
        self.m, self.k = int(mG), int(kG)
        self.Mt = make_matrix(engine, *Mt_host)
        fullG = make_matrix(engine, *Mgt_host)
        flops = engine.row_flops(fullG, self.Mt)
        del fullG
        prefix = np.zeros(self.m + 1, dtype=np.int64)
        np.cumsum(flops, out=prefix[1:])
        self.ends = equal_partition64(prefix, self.world)
        self.r0, self.r1 = int(self.ends[self.rank]), int(self.ends[self.rank + 1])
        rpG = np.asarray(rpG)
        lo, hi = int(rpG[self.r0]), int(rpG[self.r1])
        self.Mgt_local = make_matrix(engine, (rpG[self.r0:self.r1 + 1] - lo).astype(np.int32), np.asarray(ciG)[lo:hi],
                                     np.asarray(vG)[lo:hi], self.r1 - self.r0, self.k)

    def iterate(self, maxIter):
        """Runs maxIter steps; returns the replicated Mt as a dict of device tensors (rowPtr, colInd, values, ...)."""
        eng, G, me = self.engine, self.world, self.rank
        for _ in range(int(maxIter)):
            rp_loc, ci_loc, v_loc = eng.expand_prune(self.Mgt_local, self.Mt)
            nn = int(ci_loc.numel())
            if G == 1:
                rowPtr, JC, Cv, total = rp_loc, ci_loc, v_loc, nn
            else:
                offs = _segment_offsets(self.group, G, nn, rp_loc.device)
                total = int(offs[G])
                o0 = int(offs[me])
                rowPtr = eng.empty(self.m + 1, torch.int32)
                JC = eng.empty(max(total, 1), torch.int32)
                Cv = eng.empty(max(total, 1), torch.float32)
                rowPtr[self.r0:self.r1] = rp_loc[:-1] + o0
                if me == G - 1:
                    rowPtr[self.m] = total
                if nn:
                    JC[o0:o0 + nn] = ci_loc
                    Cv[o0:o0 + nn] = v_loc
                _allgatherv_csr(self.group, me, G, self.ends, self.m, offs, rowPtr, JC, Cv)
                JC, Cv = JC[:total], Cv[:total]
            self.Mt = {"rowPtr": rowPtr, "colInd": JC, "values": Cv, "rows": self.m, "cols": self.Mt["cols"],
                       "nnz": total}
        return self.Mt

    def result_host(self):
        M = self.Mt
        return (M["rowPtr"].cpu().numpy(), M["colInd"][:M["nnz"]].cpu().numpy(), M["values"][:M["nnz"]].cpu().numpy())
