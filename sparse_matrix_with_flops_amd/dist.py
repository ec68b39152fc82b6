"""Row-sharded multi-GPU SpGEMM: A split into flops-balanced contiguous row blocks, B replicated,
C's row segments exchanged with an allgatherv over RCCL/xGMI (one process per GPU).

Two implementations live here.  (1) `library_group` / `LibraryShardedSpGEMM` / `library_rmcl`: thin callers of the C ABI's
own multi-GPU entry points (include/spgemm_hip.h "multi-GPU", csrc/sharded.hpp) -- torch.distributed only carries the
128-byte RCCL id from rank 0 to the others; partition, per-shard pipeline, size exchange and the grouped ncclSend/ncclRecv
all run inside libspgemm_hip.so.  This is what bench.py --gpus N runs.  (2) `ShardedSpGEMM` / `ShardedRMCL`: the same
control flow written at the Python level over torch.distributed collectives with a pluggable local engine; it runs on
gloo with the CPU oracle as engine (tests/test_dist_gloo.py: the N>1 logic on a box without GPUs) and is bench.py's
fallback when the library's group cannot be made.

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


def footprint_prefix(row_flops, row_counts, row_nnzA):
    """The reference's alternative load measure (static scheduler, nlibs/static_omp_csr_kernel.cc:28-62
    footPrintsCrowiCount): per row of C (flops + nnz(C row) + 32 + nnz(A row)) >> 1 -- work of the numeric pass plus
    what the row writes -- 0 for an empty A row.  Returns the exclusive prefix [m+1] (int64), ready for
    equal_partition64 (the reference cuts it with arrayEqualPartition, the int32 twin of the same rule)."""
    fl = np.asarray(row_flops, dtype=np.int64)
    na = np.asarray(row_nnzA, dtype=np.int64)
    fp = np.where(na > 0, (fl + np.asarray(row_counts, dtype=np.int64) + 32 + na) >> 1, 0)
    out = np.zeros(len(fl) + 1, dtype=np.int64)
    np.cumsum(fp, out=out[1:])
    return out


def _ptr(t):
    return t.data_ptr() if t.numel() else 0


class HipEngine:
    """Local compute on one GPU through the C ABI (no CPU fallback)."""

    def __init__(self, device_index, handles=1):
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        # a handle keeps ONE pending symbolic phase; the sub-blocks of a sharded step need one each
        self.handles = [hs.Handle(device_index) for _ in range(max(1, int(handles)))]
        self.handle = self.handles[0]
        self._last = None

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

    def symbolic(self, A, B, slot=0):
        IC = self.empty(A["rows"] + 1, torch.int32)
        self.sync()
        nnz = hs.spgemm_symbolic_raw(self.handles[slot], _ptr(A["rowPtr"]), _ptr(A["colInd"]), A["nnz"], _ptr(B["rowPtr"]),
                                     _ptr(B["colInd"]), B["nnz"], A["rows"], A["cols"], B["cols"], _ptr(IC))
        return IC, nnz

    def numeric(self, A, B, IC, JC_out, C_out, slot=0):
        self.sync()
        hs.spgemm_numeric_raw(self.handles[slot], _ptr(A["rowPtr"]), _ptr(A["colInd"]), _ptr(A["values"]), A["nnz"],
                              _ptr(B["rowPtr"]), _ptr(B["colInd"]), _ptr(B["values"]), B["nnz"], A["rows"], A["cols"],
                              B["cols"], _ptr(IC), _ptr(JC_out), _ptr(C_out))

    def spmm(self, A, B):
        """One-shot C = A*B (hip_gpuSpMM: classify, symbolic, scan, numeric with no host round trip in between).
        Returns a DeviceCSR that owns the library's output arrays (released when dropped)."""
        self.sync()
        ic, jc, cv, nnz = hs.gpu_spmm_raw(self.handle, _ptr(A["rowPtr"]), _ptr(A["colInd"]), _ptr(A["values"]), A["nnz"],
                                          _ptr(B["rowPtr"]), _ptr(B["colInd"]), _ptr(B["values"]), B["nnz"],
                                          A["rows"], A["cols"], B["cols"])
        return DeviceCSR(ic, jc, cv, A["rows"], B["cols"], nnz)

    def expand_prune(self, A, B, fused=True):
        """One R-MCL step on this rank's rows, all on the device: prune(A*B).  fused: hip_rmcl_expand_prune (the row
        rule applied inside the numeric kernels, the product never reaches HBM); otherwise hip_gpuSpMM followed by
        hip_rmcl_prune.  Returns torch tensors (rowPtr[rows+1], colInd, values)."""
        self.sync()
        args = (_ptr(A["rowPtr"]), _ptr(A["colInd"]), _ptr(A["values"]), A["nnz"],
                _ptr(B["rowPtr"]), _ptr(B["colInd"]), _ptr(B["values"]), B["nnz"], A["rows"], A["cols"], B["cols"])
        if fused:
            pi, pj, pv, nn = hs.rmcl_expand_prune_raw(self.handle, *args)
        else:
            ic, jc, cv, nnzc = hs.gpu_spmm_raw(self.handle, *args)
            try:
                pi, pj, pv, nn = hs.rmcl_prune_raw(self.handle, A["rows"], ic, jc, cv, nnz=nnzc)
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


class DeviceCSR:
    """C arrays allocated by the library (spgemm_hip_malloc pool); freed back to the pool on release()/GC."""

    def __init__(self, ic, jc, cv, rows, cols, nnz):
        self.ic, self.jc, self.cv, self.rows, self.cols, self.nnz = ic, jc, cv, rows, cols, nnz

    def to_host(self):
        return (hs.d2h(self.ic, self.rows + 1, np.int32), hs.d2h(self.jc, self.nnz, np.int32),
                hs.d2h(self.cv, self.nnz, np.float32))

    def release(self):
        for p in (self.ic, self.jc, self.cv):
            if p:
                hs.dev_free(p)
        self.ic = self.jc = self.cv = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def make_matrix(engine, rowPtr, colInd, values, rows, cols):
    return {"rowPtr": engine.tensor(rowPtr, torch.int32), "colInd": engine.tensor(colInd, torch.int32),
            "values": engine.tensor(values, torch.float32), "rows": int(rows), "cols": int(cols),
            "nnz": int(rowPtr[-1]) if len(rowPtr) else 0}


def library_group(device_index, group=None):
    """This process's one-shard spgemm_group (hs.Group.of_rank): rank 0 makes the RCCL id inside the library, the id
    travels over the torch.distributed group, every rank then creates its shard.  Every rank first checks that the
    library can load RCCL at all (a rank that cannot must not leave the others waiting inside ncclCommInitRank)."""
    world = dist.get_world_size(group) if (dist and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if (dist and dist.is_initialized()) else 0
    ok = torch.tensor([0 if hs.rccl_available() else 1], dtype=torch.int32,
                      device=("cuda" if world > 1 and dist.get_backend(group) == "nccl" else "cpu"))
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MAX, group=group)
    if int(ok.item()) != 0:
        raise hs.SpgemmError("librccl could not be loaded on every rank")
    ident = hs.unique_id() if rank == 0 else bytes(hs.UNIQUE_ID_BYTES)
    if world > 1:
        t = torch.tensor(list(ident), dtype=torch.uint8, device=ok.device)
        dist.broadcast(t, 0, group=group)
        ident = bytes(t.cpu().tolist())
    return hs.Group.of_rank(world, rank, device_index, ident)


class LibraryShardedSpGEMM:
    """C = A * B row-sharded over the ranks, entirely behind the C ABI (hip_sharded_spmm_*).  A_host / B_host:
    (rowPtr, colInd, values, rows, cols) numpy tuples, identical on every rank; B_host=None means C = A*A."""

    def __init__(self, lib_group, A_host, B_host=None):
        self.group = lib_group
        self.hA = hs.CSR.from_arrays(*A_host)
        self.hB = hs.CSR.from_arrays(*B_host) if B_host is not None else None
        self.job = hs.ShardedSpMM(lib_group, self.hA, self.hB)
        self.handle = self.job.handle(0)
        self.nnz, self.total_flops = 0, None

    def step(self, gather=True):
        """one hot-path pass; -> nnz of the gathered C (gather=False: of this rank's block)"""
        self.nnz, self.total_flops = self.job.step(gather)
        return self.nnz

    def result_host(self):
        """what this rank holds after the last step: the whole C (gathered) or its own block, as an hs.CSR"""
        return self.job.result(0)

    def info(self):
        return self.job.info()


def library_rmcl(lib_group, maxIter, Mgt_host, Mt_host):
    """gpuRmclIter over the ranks behind the C ABI (hip_gpuRmclIter_sharded); every rank gets the whole result."""
    return hs.gpuRmclIter_sharded(lib_group, maxIter, hs.CSR.from_arrays(*Mgt_host), hs.CSR.from_arrays(*Mt_host))


class ShardedSpGEMM:
    """C = A * B with A row-sharded over the process group.  Every rank ends up with the whole C.

    A rank's row block is cut once more into `chunks` flops-balanced sub-blocks.  Per step: symbolic of every sub-block
    (sizes) -> one all-gather of the sizes -> numeric of sub-block k straight into its slice of the gathered arrays, and
    while sub-block k+1 computes, the send/recv pairs of sub-block k are in flight (RCCL runs on its own stream): the
    exchange hides behind the compute instead of following it.  The gathered buffers are kept across steps."""

    def __init__(self, engine, A_host, B_host=None, group=None, chunks=1, partition="flops"):
        """A_host/B_host: (rowPtr, colInd, values, rows, cols) numpy tuples, identical on every rank.
        B_host=None means C = A*A (B is the replicated full A).  partition: "flops" (arrayEqualPartition64) or
        "footprint" (flops + output-size estimate, static_omp_csr_kernel.cc:28-95 -> footprint_partition)."""
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
        self.prefix = prefix
        rpA = np.asarray(rpA)
        self.partition = partition
        if partition == "footprint":
            # one symbolic pass over the whole A (every rank, once): row lengths of C -> the reference's footprint measure
            IC_all, _ = engine.symbolic(fullA, self.B)
            counts = np.diff(IC_all.cpu().numpy().astype(np.int64))
            self.cut_prefix = footprint_prefix(flops, counts, np.diff(rpA))
        elif partition == "flops":
            self.cut_prefix = prefix
        else:
            raise ValueError(f"unknown partition {partition!r}")
        self.ends = equal_partition64(self.cut_prefix, self.world)
        r0, r1 = int(self.ends[self.rank]), int(self.ends[self.rank + 1])
        self.r0, self.r1 = r0, r1
        self.local_flops = int(prefix[r1] - prefix[r0])
        ciA, vA = np.asarray(ciA), np.asarray(vA)
        lo, hi = int(rpA[r0]), int(rpA[r1])
        self.A_local = make_matrix(engine, (rpA[r0:r1 + 1] - lo).astype(np.int32), ciA[lo:hi], vA[lo:hi], r1 - r0, self.k)
        # sub-blocks of every rank (all ranks know all cuts: the receive side needs the row ranges of its peers)
        self.chunks = max(1, min(int(chunks), len(getattr(engine, "handles", [None]))))
        self.sub_ends = []                                   # per rank: chunks+1 global row indices
        for r in range(self.world):
            a0, a1 = int(self.ends[r]), int(self.ends[r + 1])
            cut = equal_partition64(self.cut_prefix[a0:a1 + 1] - self.cut_prefix[a0], self.chunks) + a0 if a1 > a0 else \
                np.full(self.chunks + 1, a0, dtype=np.int64)
            self.sub_ends.append(cut)
        self.A_sub = []
        if self.world > 1 and self.chunks > 1:
            for c in range(self.chunks):
                s0, s1 = int(self.sub_ends[self.rank][c]), int(self.sub_ends[self.rank][c + 1])
                lo, hi = int(rpA[s0]), int(rpA[s1])
                self.A_sub.append(make_matrix(engine, (rpA[s0:s1 + 1] - lo).astype(np.int32), ciA[lo:hi], vA[lo:hi],
                                              s1 - s0, self.k))
        else:
            self.chunks = 1
            self.sub_ends = [np.array([int(self.ends[r]), int(self.ends[r + 1])], dtype=np.int64) for r in range(self.world)]
            self.A_sub = [self.A_local]
        self._buf = None                                     # gathered rowPtr / colInd / values, reused across steps
        del fullA

    # ---- one hot-path pass --------------------------------------------------------------------
    def step(self, gather=True):
        """gather=True: every rank returns the whole C (rowPtr, colInd, values).  gather=False: C stays row-sharded
        like A -- returns this rank's (local rowPtr, colInd, values) and no collective runs."""
        eng, G, me, K = self.engine, self.world, self.rank, self.chunks
        if G == 1 or not gather:
            if hasattr(eng, "spmm"):                          # one-shot entry point: no host round trip between the phases
                self._last = None                            # the consumer is done with the previous C: back to the pool
                out = eng.spmm(self.A_local, self.B)
                self._last = out
                return out
            IC_loc, nnz_loc = eng.symbolic(self.A_local, self.B)
            JC = eng.empty(max(nnz_loc, 1), torch.int32)
            Cv = eng.empty(max(nnz_loc, 1), torch.float32)
            eng.numeric(self.A_local, self.B, IC_loc, JC, Cv)
            return IC_loc, JC[:nnz_loc], Cv[:nnz_loc]
        # (1) symbolic of every sub-block -> sizes of all (rank, sub-block) segments
        ICs, sizes = [], []
        for c in range(K):
            ic, nz = eng.symbolic(self.A_sub[c], self.B, slot=c)
            ICs.append(ic)
            sizes.append(nz)
        seg = _segment_sizes(self.group, G, sizes, ICs[0].device)          # [G, K] int64
        offs = np.zeros(G * K + 1, dtype=np.int64)
        np.cumsum(seg.reshape(-1), out=offs[1:])
        total = int(offs[-1])
        if total > 0x7fffffff:
            raise hs.SpgemmError(f"nnz={total} does not fit the int32 CSR of the boundary")
        rowPtr, JC, Cv = self._buffers(total)
        # (2) numeric of sub-block c into its slice; (3) its send/recv pairs fly while sub-block c+1 computes
        reqs = []
        for c in range(K):
            o0, o1 = int(offs[me * K + c]), int(offs[me * K + c + 1])
            s0, s1 = int(self.sub_ends[me][c]), int(self.sub_ends[me][c + 1])
            eng.numeric(self.A_sub[c], self.B, ICs[c], JC[o0:o1] if o1 > o0 else JC[0:0], Cv[o0:o1] if o1 > o0 else Cv[0:0], slot=c)
            if s1 > s0:
                rowPtr[s0:s1] = ICs[c][:-1] + o0
            if me == G - 1 and c == K - 1:
                rowPtr[self.m] = total
            reqs += _exchange_block(self.group, me, G, K, c, self.sub_ends, self.m, offs, rowPtr, JC, Cv)
        for req in reqs:
            req.wait()
        return rowPtr, JC[:total], Cv[:total]

    def _buffers(self, total):
        b = self._buf
        if b is None or b[1].numel() < max(total, 1):
            cap = max(total + total // 16, 1)
            b = (self.engine.empty(self.m + 1, torch.int32), self.engine.empty(cap, torch.int32),
                 self.engine.empty(cap, torch.float32))
            self._buf = b
        return b


def _exchange_block(group, me, G, K, c, sub_ends, m, offs, rowPtr, JC, Cv):
    """Sub-block c of every rank: this rank sends its (rowPtr rows, colInd, values) segment to every peer and receives
    theirs, as grouped isend/irecv pairs (one xGMI link per peer).  Returns the pending requests."""
    ops = []
    s0, s1 = int(sub_ends[me][c]), int(sub_ends[me][c + 1])
    my_hi = s1 + (1 if (me == G - 1 and c == K - 1) else 0)
    o0, o1 = int(offs[me * K + c]), int(offs[me * K + c + 1])
    for r in range(G):
        if r == me:
            continue
        a0, a1 = int(offs[r * K + c]), int(offs[r * K + c + 1])
        rr0, rr1 = int(sub_ends[r][c]), int(sub_ends[r][c + 1]) + (1 if (r == G - 1 and c == K - 1) else 0)
        if my_hi > s0:
            ops.append(dist.P2POp(dist.isend, rowPtr[s0:my_hi], r, group))
        if rr1 > rr0:
            ops.append(dist.P2POp(dist.irecv, rowPtr[rr0:rr1], r, group))
        if o1 > o0:
            ops.append(dist.P2POp(dist.isend, JC[o0:o1], r, group))
            ops.append(dist.P2POp(dist.isend, Cv[o0:o1], r, group))
        if a1 > a0:
            ops.append(dist.P2POp(dist.irecv, JC[a0:a1], r, group))
            ops.append(dist.P2POp(dist.irecv, Cv[a0:a1], r, group))
    return dist.batch_isend_irecv(ops) if ops else []


def _segment_sizes(group, G, sizes, device):
    """all-gather of every rank's per-sub-block segment sizes -> int64 numpy [G, K]"""
    K = len(sizes)
    mine = torch.tensor(sizes, dtype=torch.int64, device=device)
    out = [torch.zeros(K, dtype=torch.int64, device=device) for _ in range(G)]
    dist.all_gather(out, mine, group=group)
    return np.stack([x.cpu().numpy() for x in out]).astype(np.int64)


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
