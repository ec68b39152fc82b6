"""Per-kernel averages of the counters collected by tools/pmc_sq.sh (gpurun_out/pmc_sq/p*/**/*counter_collection.csv)."""
import collections, csv, glob, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_sq"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
        if not m: continue
        k = m.group(1) + (m.group(2) or "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    c = {x: tot[k][x] / max(len(n[k][x]), 1) for x in tot[k]}
    if c.get("SQ_WAVE_CYCLES", 0) < 1e7: continue
    g = lambda x: c.get(x, float("nan"))
    wc = g('SQ_WAVE_CYCLES')
    print(f"{k:26s} wave_cyc={wc:.3g} wait_any={g('SQ_WAIT_ANY')/wc:.2f} wait_inst_any={g('SQ_WAIT_INST_ANY')/wc:.2f} active_inst_any={g('SQ_ACTIVE_INST_ANY')/wc:.2f}"
          f" busy_cyc={g('SQ_BUSY_CYCLES'):.3g} waves={g('SQ_WAVES'):.3g} | insts valu={g('SQ_INSTS_VALU'):.3g} salu={g('SQ_INSTS_SALU'):.3g} lds={g('SQ_INSTS_LDS'):.3g} vmem_rd={g('SQ_INSTS_VMEM_RD'):.3g} vmem_wr={g('SQ_INSTS_VMEM_WR'):.3g} smem={g('SQ_INSTS_SMEM'):.3g}"
          f" | active valu={g('SQ_ACTIVE_INST_VALU'):.3g} lds={g('SQ_ACTIVE_INST_LDS'):.3g} vmem={g('SQ_ACTIVE_INST_VMEM'):.3g} sca={g('SQ_ACTIVE_INST_SCA'):.3g}"
          f" | wait_inst_lds={g('SQ_WAIT_INST_LDS'):.3g} bank_conflict={g('SQ_LDS_BANK_CONFLICT'):.3g} lds_idx_active={g('SQ_LDS_IDX_ACTIVE'):.3g} busy_cu={g('SQ_BUSY_CU_CYCLES'):.3g} thread_cyc_valu={g('SQ_THREAD_CYCLES_VALU'):.3g}")
