#!/usr/bin/env python3
"""bench.py — the SpGEMM hot path on MI355X, measured per the driver contract.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

N > 1 without a launcher (WORLD_SIZE unset): this process starts the N ranks itself as a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` BEFORE it touches the GPU, waits, and relays
rank 0's JSON line (exit code = the child's).  Under a launcher (WORLD_SIZE set) the world size must equal --gpus.

A "step" is one pass of the hot path — per-row flop count, row binning, symbolic, scan, numeric (and for N>1 the
allgatherv of C) — over one synthetic matrix that is already resident in HBM when the timed region starts.
Metric (BASELINE.json): SpGEMM GFLOP/s = 2*P / t with P = intermediate products; output nnz/s is reported next to
it.  Rank 0 prints ONE JSON line.

Workloads (SURVEY.md §8d generator, sparse_matrix_with_flops_amd/synth.py):
  synth_1m_16    1 048 576^2, ~16 nnz/row, seed 43   <- default: the configuration the metric is quoted on
  synth_256k_16  262 144^2,  ~16 nnz/row, seed 42    (BASELINE.json configs[1])
  synth_1m_32    1 048 576^2, ~32 nnz/row, seed 44   (configs[3], the row-sharded multi-GPU case)
  web_google_surrogate  916 428^2, ~5.5 nnz/row, nnzC/P ~0.49 (configs[2] by shape; the real file is not available)
"""
import argparse
import json
import os
import subprocess
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
    # BASELINE.json configs[2] by shape (the real file is in neither container): synth.webgraph_csr, nnzC/P = 0.49
    "web_google_surrogate": dict(m=916428, seed=46, gen="web",
                                 desc="web-Google-shaped surrogate 916428^2, ~5.5 nnz/row, nnzC/P~0.49, seed 46, C=A*A "
                                      "(surrogate; reference totals unpinned)"),
}

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALL_KERNELS = 0xFFFFF


def bin_of(f):
    edges = np.array([0, 1, 4, 16, 64, 512, 2048, 4096], dtype=np.int64)   # bin b holds flops <= edges[b]; last bin beyond
    return np.searchsorted(edges, f, side="left").astype(np.int64)


# which bins each kernel covers, and whether it is a symbolic (keys only) or numeric launch
KERNEL_BINS = {
    "k_sym_g16<32,1>": ((2, 3), "sym"), "k_sym_g16": ((4,), "sym"), "k_sym_hash<1,1024>": ((5,), "sym"),
    "k_sym_hash<4,4096>": ((6,), "sym"), "k_sym_hash<8,8192>": ((7,), "sym"), "k_sym_big": ((8,), "sym"),
    "k_num_g16<32,1>": ((1, 2, 3), "num"), "k_num_g16": ((4,), "num"), "k_num_hash<1,1024>": ((5,), "num"),
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


def self_launch(args):
    """--gpus N > 1 from a plain shell: start the ranks as a child torchrun (this process has made no GPU call)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:] + "\n" + p.stderr[-8000:])
        raise SystemExit(p.returncode if p.returncode != 0 else f"the {args.gpus}-rank run printed no result line")
    got = json.loads(line)
    if got.get("n_gpus") != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group had {got.get('n_gpus')} ranks")
    print(line)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synth_1m_16", choices=sorted(WORKLOADS))
    ap.add_argument("--no-verify", action="store_true", help="skip the parity gate against the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-array entry point (hip_CSR_SpMM) timing")
    ap.add_argument("--no-gather", action="store_true",
                    help="N>1: leave C row-sharded (no allgatherv inside the timed step)")
    ap.add_argument("--chunks", type=int, default=4, help="N>1: sub-blocks per rank whose exchange overlaps the next one's numeric phase")
    ap.add_argument("--traffic-json", default=None, help="profiles/*.json with PMC-derived HBM bytes per kernel")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank).  gloo: rehearsal of the N>1 path on a box with fewer GPUs "
                         "than ranks (ranks share devices; transport over host memory; not a performance number)")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="keep the HIP events of every kernel inside the timed region (diagnostic; costs ~3 %% of a step)")
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))   # cpu_baseline: all host threads

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0:
        if args.gpus > 1:
            self_launch(args)                            # never returns
        world = 1
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    from sparse_matrix_with_flops_amd import synth
    from sparse_matrix_with_flops_amd import hipspgemm as hs
    from sparse_matrix_with_flops_amd.dist import DeviceCSR, HipEngine, ShardedSpGEMM

    if not torch.cuda.is_available() or hs.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP SpGEMM path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    wl = WORKLOADS[args.workload]
    t0 = time.time()
    if wl.get("gen") == "web":
        rp, ci, v = synth.webgraph_csr(wl["m"], wl["seed"])
    else:
        rp, ci, v = synth.powerlaw_csr(wl["m"], wl["seed"], wl["base"])
    m = wl["m"]
    gen_s = time.time() - t0
    chunks = max(1, args.chunks) if world > 1 else 1
    engine = HipEngine(local_rank, handles=chunks)
    engine.handle.selftest()
    job = ShardedSpGEMM(engine, (rp, ci, v, m, m), None, chunks=chunks)
    P = job.total_flops

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def set_timing(mask):
        for hnd in engine.handles:
            hnd.set_kernel_timing(mask)

    gather = not args.no_gather
    out = None
    for _ in range(args.warmup):
        out = None                                     # the consumer is done with the previous C: its arrays go back
        out = job.step(gather)                         # to the caching allocator and the next step reuses them

    # ---- untimed profiling pass: every kernel bracketed by HIP events -> per-kernel averages, dominant kernel
    set_timing(ALL_KERNELS)
    prof_ms, nprof = {}, 3
    for _ in range(nprof):
        out = None
        out = job.step(gather)
        for hnd in engine.handles:
            for kname, ms in hnd.stats()["ms_kernel"].items():
                prof_ms[kname] = prof_ms.get(kname, 0.0) + ms
    prof_avg = {k_: v_ / nprof for k_, v_ in prof_ms.items()}
    cand = {k_: v_ for k_, v_ in prof_avg.items() if k_ in KERNEL_BINS}
    dom = max(cand, key=cand.get) if cand else None
    dom_id = next((i for i in range(hs.NKERNELS) if hs.kernel_name(i) == dom), None) if dom else None
    set_timing(ALL_KERNELS if args.time_all_kernels else ((1 << dom_id) if dom_id is not None else 0))

    # ---- timed region
    kern_ms = {}
    phase_ms = {"ms_classify": 0.0, "ms_symbolic": 0.0, "ms_scan_alloc": 0.0, "ms_numeric": 0.0, "ms_total": 0.0}
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        out = None
        out = job.step(gather)
        for hnd in engine.handles[:chunks]:
            st = hnd.stats()                           # HIP-event durations of this step's launches (handle's stream)
            dev_ms += st["ms_total"]
            for kname, ms in st["ms_kernel"].items():
                kern_ms[kname] = kern_ms.get(kname, 0.0) + ms
            for kk in phase_ms:
                phase_ms[kk] += st[kk]
    barrier()
    elapsed = time.perf_counter() - t0
    set_timing(0)

    if isinstance(out, DeviceCSR):
        nnz_local = out.nnz
    else:
        nnz_local = int(out[1].numel())
    nnz_sum = nnz_local
    if world > 1:
        tt = torch.tensor([elapsed, dev_ms, float(nnz_local)], dtype=torch.float64,
                          device=("cuda" if args.backend == "nccl" else "cpu"))
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed, dev_ms = float(mx[0].item()), float(mx[1].item())
        nnz_sum = int(tt[2].item())
    ms_per_step = elapsed * 1e3 / args.steps
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
                                   f"allgatherv of C (send/recv pairs over xGMI) in {chunks} sub-blocks per rank, each "
                                   "overlapping the next one's numeric phase")},
        "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "backend": (args.backend if world > 1 else None),
        "output_nnz_per_s": round(nnzC / (ms_per_step * 1e-3), 1),
        # device time of the SpGEMM phases alone (max over ranks, HIP events): what the step costs without the allgatherv of C
        "compute_only": {"ms_per_step": round(dev_ms / args.steps, 4),
                         "value": round(2.0 * P / max(dev_ms / args.steps * 1e-3, 1e-12) / 1e9, 3), "unit": "GFLOP/s"},
        "gather_in_step": bool(world > 1 and gather),
        "pipeline_bytes_alg_GBs": round(bytes_alg / (ms_per_step * 1e-3) / 1e9, 2),
        "pipeline_frac_of_hbm_peak": round(bytes_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4),
    }

    if rank == 0:
        # ---- C of the last step on the host (parity gate; per-bin counts for the roofline)
        if isinstance(out, DeviceCSR):
            rpc, jc_h, cv_h = out.to_host()
        else:
            rpc, jc_h, cv_h = (x.cpu().numpy() for x in out)
        rpc = rpc.astype(np.int64)

        # ---- roofline of the dominant kernel (rank 0's local rows; duration from HIP events INSIDE the timed region)
        flops_rows = engine.row_flops(job.A_local, job.B)
        rpl = job.A_local["rowPtr"].cpu().numpy().astype(np.int64)
        cnt_rows = np.diff(rpc[job.r0:job.r1 + 1]) if (world > 1 and gather) else np.diff(rpc)
        b = bin_of(flops_rows)
        per_bin = {}
        for q in range(9):
            sel = b == q
            per_bin[q] = (int(sel.sum()), int(np.diff(rpl)[sel].sum()), int(flops_rows[sel].sum()), int(cnt_rows[sel].sum()))
        roof = None
        if dom and dom in kern_ms:
            avg_dom = kern_ms[dom] / args.steps
            bins, kind = KERNEL_BINS[dom]
            rows_ = sum(per_bin[q][0] for q in bins)
            nza_ = sum(per_bin[q][1] for q in bins)
            p_ = sum(per_bin[q][2] for q in bins)
            nzc_ = sum(per_bin[q][3] for q in bins)
            ab = algorithmic_bytes(kind, rows_, nza_, p_, nzc_)
            ach = ab / (avg_dom * 1e-3) / 1e9
            traffic, tsrc = None, None
            tj = args.traffic_json or os.path.join(ROOT, "profiles", f"r02_{args.workload}_traffic.json")
            if os.path.exists(tj):      # PMC-derived HBM bytes per launch, collected by profiles/collect.sh (separate passes)
                traffic = json.load(open(tj)).get("kernels", {}).get(dom, {}).get("hbm_bytes_raw")
                tsrc = (f"{os.path.relpath(tj, ROOT)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, collected "
                        "separately (not in this run)")
            roof = {"bound": "hbm", "kernel": dom, "avg_launch_ms": round(avg_dom, 4), "alg_bytes_per_launch": ab,
                    "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": tsrc,
                    "rows": rows_, "products": p_, "nnzC": nzc_,
                    "timing": "dominant kernel: HIP events on the handle's stream inside the timed region; all_kernels_avg_ms: "
                              f"a separate untimed pass of {nprof} steps with every kernel bracketed",
                    "all_kernels_avg_ms": {k_: round(v_, 4) for k_, v_ in sorted(prof_avg.items(), key=lambda kv: -kv[1])},
                    "phases_avg_ms": {k_: round(v_ / args.steps, 4) for k_, v_ in phase_ms.items()}}
        result["roofline"] = roof

        # ---- parity gate (every N) + CPU baseline (N=1 only); the oracle is the checker, never the thing measured above
        from oracle import pyoracle as po
        A = po.CSRHost(rp, ci, v, m, m)
        if world == 1 and not args.no_cpu_baseline:
            ncpu = os.cpu_count() or 1
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
                           f"OpenMP threads={int(threads)} of host cpus={ncpu}"),
                "ms": round(best * 1e3, 2)}
        if world == 1 and not args.no_host_api:
            # the drop-in a reference caller of CSR::*spmm gets (nlibs/CSR.cc:122-134): host arrays in, malloc()ed host
            # arrays out.  PCIe-inclusive, never `value`.
            out = None
            hA = hs.CSR.from_arrays(rp, ci, v, m, m)
            runs = hs.host_api_timed(hA, hA, reps=3)
            best = min(runs[1:], key=lambda r_: r_["ms_total"])          # first run: pinned slots are allocated
            result["host_api"] = {
                "entry": "hip_CSR_SpMM (host CSR in, malloc()ed host CSR out)", "ms": round(best["ms_total"], 2),
                "ms_h2d": round(best["ms_h2d"], 2), "ms_device": round(best["ms_device"], 2), "ms_d2h": round(best["ms_d2h"], 2),
                "GB_h2d": round(best["bytes_h2d"] / 1e9, 3), "GB_d2h": round(best["bytes_d2h"] / 1e9, 3),
                "d2h_GBs": round(best["bytes_d2h"] / max(best["ms_d2h"], 1e-9) / 1e6, 1),
                "GFLOPs_incl_pcie": round(2.0 * P / (best["ms_total"] * 1e-3) / 1e9, 2), "runs_ms": [round(r_["ms_total"], 1) for r_ in runs]}
        if not args.no_verify and (world == 1 or gather):
            want = po.omp_spmm(A, A)
            got = po.CSRHost(rpc, jc_h, cv_h, m, m)
            ok = np.array_equal(got.rowPtr, want.rowPtr)
            if ok:
                g2, w2 = got.canonical(), want.canonical()
                ok = np.array_equal(g2.colInd, w2.colInd)
                if ok:
                    a_, b_ = g2.values.astype(np.float64), w2.values.astype(np.float64)
                    ok = bool(np.all(np.abs(a_ - b_) <= 1e-6 * np.maximum(np.abs(a_), np.abs(b_))))
            result["parity"] = ("ok (rowPtr, sorted colInd bit-exact; values rel<=1e-6 vs CPU oracle"
                                + (f"; gathered C of {world} ranks checked on rank 0)" if world > 1 else ")")) if ok else "FAILED"
            if not ok:
                print(json.dumps(result))
                raise SystemExit("parity gate failed")
        result["setup"] = {"generate_s": round(gen_s, 2), "host_cpus": os.cpu_count()}
        print(json.dumps(result))
    out = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
