#!/bin/bash
# L2 / fabric-side counters for one workload (separate --pmc passes of <= 4 TCC counters, kernel-trace only)
WL=${1:-synth_1m_16}
OUT=$PWD/gpurun_out/pmc_tcc; rm -rf $OUT; mkdir -p $OUT
ARGS="$PWD/bench.py --workload $WL --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-host-api --no-other-workloads"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_BUSY_avr TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1; echo "pass $i exit=$?"
done
cd - >/dev/null
python3 - <<'PY'
import collections, csv, glob, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmc_tcc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
        if not m: continue
        k = m.group(1) + (m.group(2) or "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("gpurun_out/tcc_summary.txt", "w") as out:
    for k in sorted(tot):
        c = {x: tot[k][x] / max(len(n[k][x]), 1) for x in tot[k]}
        if c.get("TCC_REQ_sum", 0) < 1e5: continue
        g = lambda x: c.get(x, float("nan"))
        rd_bytes = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
        line = (f"{k:26s} EA_RD={g('TCC_EA0_RDREQ_sum'):.3g} (32B {g('TCC_EA0_RDREQ_32B_sum'):.3g} 64B {g('TCC_EA0_RDREQ_64B_sum'):.3g} 128B {g('TCC_EA0_RDREQ_128B_sum'):.3g}) "
                f"rd_bytes~{rd_bytes/1e6:.0f}MB RD_DRAM={g('TCC_EA0_RDREQ_DRAM_sum'):.3g} EA_WR={g('TCC_EA0_WRREQ_sum'):.3g} (64B {g('TCC_EA0_WRREQ_64B_sum'):.3g}) WR_DRAM={g('TCC_EA0_WRREQ_DRAM_sum'):.3g} "
                f"| L2 hit={g('TCC_HIT_sum'):.3g} miss={g('TCC_MISS_sum'):.3g} req={g('TCC_REQ_sum'):.3g} read={g('TCC_READ_sum'):.3g} "
                f"| TCP->TCC rd={g('TCP_TCC_READ_REQ_sum'):.3g} lat={g('TCP_TCC_READ_REQ_LATENCY_sum')/max(g('TCP_TCC_READ_REQ_sum'),1):.0f} wr={g('TCP_TCC_WRITE_REQ_sum'):.3g} L1acc={g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3g} "
                f"| TCC_BUSY={g('TCC_BUSY_avr'):.3g} TA_BUSY={g('TA_BUSY_avr'):.3g} GUI={g('GRBM_GUI_ACTIVE'):.3g}")
        print(line); out.write(line + "\n")
PY
