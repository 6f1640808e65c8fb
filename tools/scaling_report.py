#!/usr/bin/env python3
"""tools/scaling_report.py — the reference's scaling table (scripts/run_benchmark.sh:54-68:
speedup, efficiency, Karp-Flatt serial fraction) from bench.py JSON lines, extended with
Mcell-updates/s, algorithmic GB/s and fraction of the HBM roofline.

  python tools/scaling_report.py bench_n1.json bench_n2.json bench_n4.json bench_n8.json > scaling.csv
Each input file holds the single JSON line bench.py printed for that GPU count."""
import json
import sys


def main(paths):
    rows = []
    for p in paths:
        with open(p) as f:
            for ln in f:
                ln = ln.strip()
                if ln.startswith("{"):
                    rows.append(json.loads(ln))
    rows.sort(key=lambda r: r["n_gpus"])
    if not rows:
        raise SystemExit("no bench lines found")
    base = rows[0]
    t1 = base["ms_per_step"] * base["n_gpus"]  # extrapolated 1-GPU time if N=1 is missing
    print("n_gpus,ms_per_step,mcell_updates_per_s,algorithmic_gb_per_s,frac_of_8tbs_x_n,speedup,efficiency,karp_flatt")
    for r in rows:
        n = r["n_gpus"]
        sp = t1 / r["ms_per_step"]
        eff = sp / n
        kf = "" if n == 1 else f"{(1.0 / sp - 1.0 / n) / (1.0 - 1.0 / n):.6f}"
        gbs = r["value"] * 1e6 * 16.0 / 1e9
        print(f"{n},{r['ms_per_step']:.6f},{r['value']:.1f},{gbs:.1f},{gbs / (8000.0 * n):.4f},{sp:.4f},{eff:.4f},{kf}")


if __name__ == "__main__":
    main(sys.argv[1:])
