#!/usr/bin/env python3
"""Condense a profiles/collect.sh run into the files that get committed:
   profiles/<tag>_<workload>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary
   profiles/<tag>_<workload>_traffic.json        per kernel: avg FETCH_SIZE / WRITE_SIZE per launch -> HBM bytes
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are reported in KiB by
rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced streams (x2 correction).  This path
gathers 4-byte elements in short runs; the TCC/EA counters (profiles/*tcc_counters*) show that every fabric read request of
these kernels is a 128-byte one, so the x2 correction applies: bench.py reports the x2 figure as roofline.traffic
(traffic_fetch_x2) and the uncorrected sum beside it (traffic_raw)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    base, tmpl = m.group(1), m.group(2) or ""
    if tmpl:
        parts = [p.strip() for p in tmpl[1:-1].split(",")]
        tmpl = "<" + ",".join(parts[:2]) + ">" if base in ("k_sym_hash", "k_num_hash", "k_sym_small", "k_num_small") else ""
    return base + tmpl


def counter_avg(sub, counter):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    files = sorted(files, key=os.path.getmtime)[-1:]      # gpurun merges runs into one directory: newest only
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if k:
                tot[k] += float(r["Counter_Value"])
                n[k].add(r["Dispatch_Id"])
    return {k: tot[k] / max(len(n[k]), 1) for k in tot}


stats = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(here, f"{tag}_{wl}_kernel_stats.csv"))
fetch, write = counter_avg("fetch", "FETCH_SIZE"), counter_avg("write", "WRITE_SIZE")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    traffic[k] = {"fetch_KiB_raw": round(f_kib, 1), "write_KiB": round(w_kib, 1),
                  "hbm_bytes_raw": int((f_kib + w_kib) * 1024), "hbm_bytes_fetch_x2": int((2 * f_kib + w_kib) * 1024)}
# provenance: the hashes of the kernel sources the profiled library was built from (its .buildinfo).  bench.py reports
# these bytes only while the library in use was built from the same kernel sources.
dev_sha = {}
try:
    lines = open(os.path.join(here, "..", "sparse_matrix_with_flops_amd", "libspgemm_hip.so.buildinfo")).read().splitlines()
    for ln in lines[lines.index("sources sha256:") + 1:]:
        h_, name = ln.split()
        if name.endswith("_device.hpp"):
            dev_sha[name] = h_
except (OSError, ValueError):
    pass
json.dump({"workload": wl, "note": "per launch averages; see header of profiles/summarize.py", "kernels": traffic,
           "device_code_sha256": dev_sha},
          open(os.path.join(here, f"{tag}_{wl}_traffic.json"), "w"), indent=1, sort_keys=True)
print("wrote", sorted(x for x in os.listdir(here) if x.startswith(tag)))
