#!/bin/bash
# After tools/evidence.sh <tag> has been merged back from the GPU box: condense gpurun_out/ into the tracked profiles/<tag>_*
# files (run locally).   tools/refresh_profiles.sh r04
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
for wl in synth_1m_16 web_google_surrogate synth_256k_16; do python3 profiles/summarize.py gpurun_out/prof_${TAG}_$wl $TAG $wl > /dev/null; done
cp "$(ls -t gpurun_out/prof_${TAG}_rmcl_500k/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_rmcl_500k_kernel_stats.csv
TAG=$TAG python3 - <<'PY'
import glob, json, os, sys
sys.path.insert(0, '.')
import bench
tag = os.environ['TAG']
# the bench lines are produced BEFORE the PMC passes of the same call: their roofline.traffic* fields are filled here from the
# traffic files of that call (same commit, same kernel sources: bench.traffic_for checks the recorded hashes)
for f in sorted(glob.glob(f'gpurun_out/{tag}_bench_*.json')):
    name = os.path.basename(f)[len(tag) + 7:-5]
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    wl = d['config'].get('name')
    if r.get('kernel') and r.get('alg_bytes_per_launch') and 'expand_prune' not in r['kernel'] and not os.environ.get('NO_TRAFFIC'):
        tr, src = bench.traffic_for(wl, r['kernel'])
        if tr:
            ab = r['alg_bytes_per_launch']
            r.update(traffic=tr['fetch_x2'], traffic_raw=tr['raw'], traffic_fetch_x2=tr['fetch_x2'],
                     traffic_over_alg=round(tr['fetch_x2'] / ab, 3),
                     traffic_source=src.replace('(not in this run)', '(same gpurun call, after this line)'))
    json.dump(d, open(f'profiles/{tag}_bench_{name}.json', 'w'))
    print(name, d['ms_per_step'], d['value'], r.get('kernel'), r.get('frac'), r.get('traffic_over_alg'))
PY
[ -f gpurun_out/${TAG}_sq_counters_1m.txt ] && cp gpurun_out/${TAG}_sq_counters_1m.txt profiles/
cat gpurun_out/${TAG}_commit.txt 2>/dev/null || true
