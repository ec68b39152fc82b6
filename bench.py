#!/usr/bin/env python3
"""bench.py — the SpGEMM hot path on MI355X, measured per the driver contract.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path — per-row flop count, row binning, symbolic, scan, numeric (and for
N>1 the allgatherv of C) — over one synthetic matrix that is already resident in HBM when the timed region
starts.  Metric (BASELINE.json): SpGEMM GFLOP/s = 2*P / t with P = intermediate products; output nnz/s is
reported next to it.  Rank 0 prints ONE JSON line.

Workloads (SURVEY.md §8d generator, sparse_matrix_with_flops_amd/synth.py):
  synth_1m_16    1 048 576^2, ~16 nnz/row, seed 43   <- default: the configuration the metric is quoted on
  synth_256k_16  262 144^2,  ~16 nnz/row, seed 42    (BASELINE.json configs[1])
  synth_1m_32    1 048 576^2, ~32 nnz/row, seed 44   (configs[3], the row-sharded multi-GPU case)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "synth_1m_16": dict(m=1 << 20, seed=43, base=2, desc="synthetic power-law CSR 1048576^2, ~16 nnz/row, seed 43, C=A*A"),
    "synth_256k_16": dict(m=1 << 18, seed=42, base=2, desc="synthetic power-law CSR 262144^2, ~16 nnz/row, seed 42, C=A*A"),
    "synth_1m_32": dict(m=1 << 20, seed=44, base=4, desc="synthetic power-law CSR 1048576^2, ~32 nnz/row, seed 44, C=A*A"),
    "synth_64k_16": dict(m=1 << 16, seed=17, base=2, desc="synthetic power-law CSR 65536^2 (smoke-sized)"),
}

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def bin_of(f):
    edges = np.array([0, 1, 4, 16, 64, 512, 2048, 4096], dtype=np.int64)   # bin b holds flops <= edges[b]; last bin beyond
    return np.searchsorted(edges, f, side="left").astype(np.int64)


# which bins each kernel covers, and whether it is a symbolic (keys only) or numeric launch
KERNEL_BINS = {
    "k_sym_small<4,32>": ((2, 3), "sym"), "k_sym_g16": ((4,), "sym"), "k_sym_hash<1,1024>": ((5,), "sym"),
    "k_sym_hash<4,4096>": ((6,), "sym"), "k_sym_hash<8,8192>": ((7,), "sym"), "k_sym_big": ((8,), "sym"),
    "k_num_small<4,32>": ((1, 2, 3), "num"), "k_num_g16": ((4,), "num"), "k_num_hash<1,1024>": ((5,), "num"),
    "k_num_hash<4,4096>": ((6,), "num"), "k_num_hash<8,8192>": ((7,), "num"), "k_num_big": ((8,), "num"),
    "k_num_bighash": ((8,), "num"),
}


def algorithmic_bytes(kind, rows, nnzA, P, nnzC):
    """Per-launch algorithmic bytes.  numeric: SURVEY.md §8(d) BYTES_ALG restricted to the rows of
    the launch = 8 rows + 16 nnzA + 8 P + 8 nnzC.  symbolic (not credited by BYTES_ALG, reported for completeness):
    rowPtr/rowIds 8 rows + A cols and two B.rowPtr reads 12 nnzA + B cols 4 P + counts 4 rows."""
    if kind == "num":
        return 8 * rows + 16 * nnzA + 8 * P + 8 * nnzC
    return 12 * rows + 12 * nnzA + 4 * P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synth_1m_16", choices=sorted(WORKLOADS))
    ap.add_argument("--no-verify", action="store_true", help="skip the parity gate against the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true",
                    help="N>1: leave C row-sharded (no allgatherv inside the timed step)")
    ap.add_argument("--traffic-json", default=None, help="profiles/*.json with PMC-derived HBM bytes per kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sparse_matrix_with_flops_amd import synth
    from sparse_matrix_with_flops_amd import hipspgemm as hs
    from sparse_matrix_with_flops_amd.dist import HipEngine, ShardedSpGEMM

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available() or hs.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP SpGEMM path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    t0 = time.time()
    rp, ci, v = synth.powerlaw_csr(wl["m"], wl["seed"], wl["base"])
    m = wl["m"]
    gen_s = time.time() - t0
    engine = HipEngine(local_rank)
    engine.handle.selftest()
    job = ShardedSpGEMM(engine, (rp, ci, v, m, m), None)
    P = job.total_flops

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    out = None
    gather = not args.no_gather
    for _ in range(args.warmup):
        out = None                                     # the consumer is done with the previous C: its arrays go back
        out = job.step(gather)                         # to the caching allocator and the next step reuses them
    kern_ms = {}
    phase_ms = {"ms_classify": 0.0, "ms_symbolic": 0.0, "ms_scan_alloc": 0.0, "ms_numeric": 0.0, "ms_total": 0.0}
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        out = None
        out = job.step(gather)
        st = engine.stats()
        dev_ms += st["ms_total"]                       # HIP-event durations of this step's launches (handle's stream)
        for kname, ms in st["ms_kernel"].items():
            kern_ms[kname] = kern_ms.get(kname, 0.0) + ms
        for kk in phase_ms:
            phase_ms[kk] += st[kk]
    barrier()
    elapsed = time.perf_counter() - t0
    nnz_local = int(out[1].numel())
    if world > 1:
        tt = torch.tensor([elapsed, dev_ms, float(nnz_local)], dtype=torch.float64, device="cuda")
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed, dev_ms = float(mx[0].item()), float(mx[1].item())
        nnz_sum = int(tt[2].item())
    ms_per_step = elapsed * 1e3 / args.steps
    rowPtrC, JC, CV = out
    nnzC = nnz_local if (world == 1 or gather) else nnz_sum
    nnzA = int(rp[-1])
    bytes_alg = synth.bytes_alg(m, nnzA, P, nnzC)
    gflops = 2.0 * P / (ms_per_step * 1e-3) / 1e9

    result = {
        "metric": "SpGEMM GFLOP/s (2*intermediate_nnz/sec), C=A*A on 1M-row CSR",
        "value": round(gflops, 3), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["desc"], "name": args.workload, "m": m, "nnzA": nnzA, "intermediate_nnz_P": P,
                   "nnzC": nnzC, "bytes_alg": bytes_alg,
                   "parallelism": ("single GPU" if world == 1 else f"A row-sharded by flops over {world} GPUs, B replicated, "
                                   "allgatherv of C (send/recv pairs over xGMI)")},
        "output_nnz_per_s": round(nnzC / (ms_per_step * 1e-3), 1),
        # device time of the SpGEMM phases alone (max over ranks, HIP events): what the step costs without the allgatherv
        # of C.  The gather moves 8*nnzC bytes into every GPU; at xGMI link rates that exceeds the compute time at any N.
        "compute_only": {"ms_per_step": round(dev_ms / args.steps, 4),
                         "value": round(2.0 * P / (dev_ms / args.steps * 1e-3) / 1e9, 3), "unit": "GFLOP/s"},
        "gather_in_step": bool(world > 1 and gather),
        "pipeline_bytes_alg_GBs": round(bytes_alg / (ms_per_step * 1e-3) / 1e9, 2),
        "pipeline_frac_of_hbm_peak": round(bytes_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4),
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (rank 0's local rows; durations from HIP events in the timed region)
        flops_rows = engine.row_flops(job.A_local, job.B)
        rpl = job.A_local["rowPtr"].cpu().numpy().astype(np.int64)
        rpc = rowPtrC.cpu().numpy().astype(np.int64)
        cnt_rows = np.diff(rpc[job.r0:job.r1 + 1]) if (world > 1 and gather) else np.diff(rpc)
        b = bin_of(flops_rows)
        per_bin = {}
        for q in range(9):
            sel = b == q
            per_bin[q] = (int(sel.sum()), int(np.diff(rpl)[sel].sum()), int(flops_rows[sel].sum()), int(cnt_rows[sel].sum()))
        avg = {k_: v_ / args.steps for k_, v_ in kern_ms.items()}
        cand = {k_: v_ for k_, v_ in avg.items() if k_ in KERNEL_BINS}
        dom = max(cand, key=cand.get) if cand else None
        roof = None
        if dom:
            bins, kind = KERNEL_BINS[dom]
            rows_ = sum(per_bin[q][0] for q in bins)
            nza_ = sum(per_bin[q][1] for q in bins)
            p_ = sum(per_bin[q][2] for q in bins)
            nzc_ = sum(per_bin[q][3] for q in bins)
            ab = algorithmic_bytes(kind, rows_, nza_, p_, nzc_)
            ach = ab / (avg[dom] * 1e-3) / 1e9
            traffic = None
            tj = args.traffic_json or os.path.join(ROOT, "profiles", f"r01_{args.workload}_traffic.json")
            if os.path.exists(tj):      # PMC-derived HBM bytes per launch, collected by profiles/collect.sh (separate passes)
                traffic = json.load(open(tj)).get("kernels", {}).get(dom, {}).get("hbm_bytes_raw")
            roof = {"bound": "hbm", "kernel": dom, "avg_launch_ms": round(avg[dom], 4), "alg_bytes_per_launch": ab,
                    "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "rows": rows_, "products": p_, "nnzC": nzc_,
                    "all_kernels_avg_ms": {k_: round(v_, 4) for k_, v_ in sorted(avg.items(), key=lambda kv: -kv[1])},
                    "phases_avg_ms": {k_: round(v_ / args.steps, 4) for k_, v_ in phase_ms.items()}}
        result["roofline"] = roof

        # ---- parity gate + CPU baseline (rank 0, N=1 only; the oracle is the checker, never the thing measured above)
        if world == 1:
            from oracle import pyoracle as po
            A = po.CSRHost(rp, ci, v, m, m)
            if not args.no_cpu_baseline:
                use_ref = po.have_ref()
                times = []
                budget_t0 = time.time()
                for i in range(5):
                    dt, _ = po.time_omp_spmm(A, A, use_ref)          # the C call alone, outputs freed unread
                    times.append(dt)
                    if time.time() - budget_t0 > 25.0 and i >= 1:
                        break
                best = float(np.median(times[1:])) if len(times) > 1 else times[0]
                threads = po.ref().ref_max_threads() if use_ref else po.lib().oracle_max_threads()
                result["cpu_baseline"] = {
                    "value": round(2.0 * P / best / 1e9, 4), "unit": "GFLOP/s", "cores": int(threads),
                    "kind": "reference" if use_ref else "port",
                    "sample": (f"{'omp_CSR_SpMM (reference sources, oracle/_ref)' if use_ref else 'oracle_omp_spmm (C restatement of omp_CSR_SpMM)'} "
                               f"on the whole {args.workload} matrix, stride 512, incl. per-thread scratch allocation as in the "
                               f"reference's 4-argument wrapper, {len(times)} runs (first = warm-up), median {best * 1e3:.1f} ms; "
                               f"host cores={os.cpu_count()}"),
                    "ms": round(best * 1e3, 2)}
                want = None
            else:
                want = None
            if not args.no_verify:
                if want is None:
                    want = po.omp_spmm(A, A)
                got = po.CSRHost(rowPtrC.cpu().numpy(), JC.cpu().numpy(), CV.cpu().numpy(), m, m)
                ok = np.array_equal(got.rowPtr, want.rowPtr)
                if ok:
                    g2, w2 = got.canonical(), want.canonical()
                    ok = np.array_equal(g2.colInd, w2.colInd)
                    if ok:
                        a_, b_ = g2.values.astype(np.float64), w2.values.astype(np.float64)
                        ok = bool(np.all(np.abs(a_ - b_) <= 1e-6 * np.maximum(np.abs(a_), np.abs(b_))))
                result["parity"] = "ok (rowPtr, sorted colInd bit-exact; values rel<=1e-6 vs CPU oracle)" if ok else "FAILED"
                if not ok:
                    print(json.dumps(result))
                    raise SystemExit("parity gate failed")
        result["setup"] = {"generate_s": round(gen_s, 2), "host_cpus": os.cpu_count()}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
