#!/usr/bin/env python3
"""tools/scaling_report.py — the reference's scaling table (scripts/run_benchmark.sh:54-68:
speedup S = T1/Tp, efficiency E = S/P, Karp-Flatt serial fraction (1/S - 1/P)/(1 - 1/P)) from bench.py
JSON lines, extended with Mcell-updates/s, the step-equivalent GB/s (16 B per cell update), the real HBM
GB/s of the whole job (PMC bytes x launches / wall time, when the line carries it) and the HBM roofline
fraction of the dominant kernel.

  python tools/scaling_report.py bench_n1.json bench_n2.json bench_n4.json bench_n8.json > scaling.csv
Each input file holds the single JSON line bench.py printed for that GPU count."""
import json
import sys

HEADER = ("n_gpus,ms_per_step,mcell_updates_per_s,step_equivalent_gb_per_s,hbm_gb_per_s_measured,"
          "kernel_hbm_roofline_frac,speedup,efficiency,karp_flatt")


def table(rows):
    rows = sorted(rows, key=lambda r: r["n_gpus"])
    if not rows:
        raise SystemExit("no bench lines found")
    base = rows[0]
    t1 = base["ms_per_step"] * base["n_gpus"]  # extrapolated 1-GPU time if N=1 is missing
    out = [HEADER]
    for r in rows:
        n = r["n_gpus"]
        sp = t1 / r["ms_per_step"]
        eff = sp / n
        kf = "" if n == 1 else f"{(1.0 / sp - 1.0 / n) / (1.0 - 1.0 / n):.6f}"
        gbs = r["value"] * 1e6 * 16.0 / 1e9
        real = (r.get("config") or {}).get("hbm_gbs_whole_job")
        frac = (r.get("roofline") or {}).get("frac")
        out.append(f"{n},{r['ms_per_step']:.6f},{r['value']:.1f},{gbs:.1f},{'' if real is None else f'{real:.1f}'},"
                   f"{'' if frac is None else f'{frac:.4f}'},{sp:.4f},{eff:.4f},{kf}")
    return out


def main(paths):
    rows = []
    for p in paths:
        with open(p) as f:
            for ln in f:
                ln = ln.strip()
                if ln.startswith("{"):
                    rows.append(json.loads(ln))
    print("\n".join(table(rows)))


if __name__ == "__main__":
    main(sys.argv[1:])
