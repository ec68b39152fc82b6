#!/bin/bash
# After tools/gpu_r3c.sh has been merged back: condense gpurun_out/ into the tracked profiles/r03_* files (run locally).
set -e
cd "$(dirname "$0")/.."
for wl in synth_1m_16 web_google_surrogate synth_256k_16; do python3 profiles/summarize.py gpurun_out/prof_r03_$wl r03 $wl > /dev/null; done
cp "$(ls -t gpurun_out/prof_r03_rmcl_500k/trace/*/*kernel_stats.csv | head -1)" profiles/r03_rmcl_500k_kernel_stats.csv
cp gpurun_out/r03_sq_counters_1m.txt gpurun_out/r03_tcc_counters_1m.txt profiles/
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
# the bench lines are produced BEFORE the PMC passes of the same call: their roofline.traffic* fields are filled from the
# traffic files of that call here
for wl in ['synth_1m_16', 'web_google_surrogate', 'synth_256k_16', 'synth_1m_32', 'rmcl_500k']:
    d = json.load(open(f'gpurun_out/r03_bench_{wl}.json'))
    r = d.get('roofline') or {}
    if wl in ('synth_1m_16', 'web_google_surrogate', 'synth_256k_16'):
        tr, src = bench.traffic_for(wl, r['kernel'])
        ab = r['alg_bytes_per_launch']
        r.update(traffic=tr['fetch_x2'], traffic_raw=tr['raw'], traffic_fetch_x2=tr['fetch_x2'],
                 traffic_over_alg=round(tr['fetch_x2'] / ab, 3),
                 traffic_source=src.replace('(not in this run)', '(same gpurun call, after this line)'))
        print(wl, r['kernel'], r['avg_launch_ms'], r['frac'], r['traffic_over_alg'])
    json.dump(d, open(f'profiles/r03_bench_{wl}.json', 'w'))
    print(wl, d['ms_per_step'], d['value'])
for a, b in [('gpurun_out/r03_bench_group_rehearsal.json', 'profiles/r03_bench_group_rehearsal_one_rank.json'),
             ('gpurun_out/r03_bench_n2_gloo.json', 'profiles/r03_bench_n2_gloo_rehearsal.json')]:
    json.dump(json.load(open(a)), open(b, 'w'))
PY
cat gpurun_out/r03_commit.txt 2>/dev/null || true
