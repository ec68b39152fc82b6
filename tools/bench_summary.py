"""Summaries of bench.py JSON lines: one file -> headline + phases + kernels; --ab A B files... -> medians per arm."""
import json
import statistics
import sys


def load(path):
    txt = open(path).read().strip().splitlines()
    return json.loads(txt[-1])


def brief(d):
    r = d.get("roofline") or {}
    ks = {k: round(v, 4) for k, v in sorted(r.get("all_kernels_avg_ms", {}).items(), key=lambda kv: -kv[1]) if v > 0.02}
    return (f"{d['config'].get('name')}: {d['ms_per_step']} ms  {d['value']} {d['unit']}  parity={str(d.get('parity'))[:10]} "
            f"pipeline_frac={d.get('pipeline_frac_of_hbm_peak')}\n  phases={r.get('phases_avg_ms')}\n  kernels={ks}")


def main():
    a = sys.argv[1:]
    if a and a[0] == "--ab":
        names = {"A": a[1], "B": a[2]}
        arms = {}
        for f in a[3:]:
            parts = f.rsplit("/", 1)[-1][3:-5].rsplit("_", 2)       # ab_<wl>_<arm>_<rep>.json
            d = load(f)
            arms.setdefault((parts[0], parts[1]), []).append(d)
        for (wl, arm), ds in sorted(arms.items()):
            ms = [d["ms_per_step"] for d in ds]
            kern = {}
            for d in ds:
                for k, v in d["roofline"].get("all_kernels_avg_ms", {}).items():
                    kern.setdefault(k, []).append(v)
            km = {k: round(statistics.median(v), 4) for k, v in kern.items() if statistics.median(v) > 0.03}
            print(f"{wl} arm {arm} [{names[arm]}]: median {statistics.median(ms):.4f} ms  runs {sorted(ms)}  "
                  f"parity {set(str(d.get('parity'))[:8] for d in ds)}\n   kernels(median) {km}")
    else:
        for f in a:
            print(brief(load(f)))


if __name__ == "__main__":
    main()
