#!/usr/bin/env python3
"""tools/isa_summary.py — per-kernel and per-loop instruction census of a gfx950 assembly listing.

  hipcc ... -S --cuda-device-only -o k.s kernels.hip -Rpass-analysis=kernel-resource-usage 2> k.res.txt
  python tools/isa_summary.py k.s k.res.txt [--kernel SUBSTR] > profiles/rNN_isa_summary.txt

For every kernel whose mangled name contains SUBSTR: VGPRs / SGPR spills / occupancy from the resource remarks, and for
every LOOP (a label that is the target of a backward branch, up to that branch) with at least --min-fp64 fp64
instructions: counts of fp64 add/mul/fma, DPP moves, v_readlane/v_writelane (SGPR spill traffic), v_cndmask (per-lane
selects), v_cmp, other VALU, scalar instructions, s_waitcnt, global loads / stores."""
import argparse
import collections
import re
import sys


def classify(ins, line):
    if ins.startswith(("v_add_f64", "v_mul_f64", "v_fma_f64")):
        return "fp64"
    if "dpp" in line or "row_" in line or "wave_sh" in line:
        return "dpp"
    if ins.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if ins.startswith("v_cndmask"):
        return "cndmask"
    if ins.startswith("v_cmp"):
        return "vcmp"
    if ins.startswith(("v_accvgpr",)):
        return "acc"
    if ins.startswith("v_"):
        return "valu_other"
    if ins.startswith("s_waitcnt"):
        return "waitcnt"
    if ins.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if ins.startswith("s_"):
        return "salu"
    if ins.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "load"
    if ins.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "store"
    if ins.startswith("ds_"):
        return "lds"
    return "other"


def kernels(path):
    name, body = None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name and ln.strip().startswith(".end_amdhsa_kernel"):
            name = None
        if name:
            if ln.strip().startswith("s_endpgm") and False:
                pass
            body.append(ln.rstrip("\n"))
            if ln.strip().startswith(".section") or ln.strip().startswith(".rodata"):
                yield name, body
                name = None


def loops(body):
    labels = {}
    ins = []  # (idx, mnemonic, line)
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        ins.append((t.split()[0], t))
    out = []
    for k, (mn, t) in enumerate(ins):
        if mn.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                out.append((tgt, labels[tgt], k))
    return ins, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("res", nargs="?")
    ap.add_argument("--kernel", default="k_sweepO_dpp")
    ap.add_argument("--min-fp64", type=int, default=200)
    args = ap.parse_args()
    res = {}
    if args.res:
        cur = None
        for ln in open(args.res):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = m.group(1)
                res[cur] = {}
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", ln)
            if m and cur and "Function Name" not in ln:
                res[cur][m.group(1).strip()] = m.group(2)
    order = ["fp64", "dpp", "lane", "cndmask", "vcmp", "valu_other", "salu", "branch", "waitcnt", "load", "store", "lds", "other"]
    for name, body in kernels(args.asm):
        if args.kernel not in name:
            continue
        r = res.get(name, {})
        print(f"== {name}")
        print("   " + "  ".join(f"{k}={r[k]}" for k in ("VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill",
                                                            "ScratchSize", "Occupancy", "LDS Size") if k in r))
        ins, lps = loops(body)
        tot = collections.Counter(classify(mn, t) for mn, t in ins)
        print("   whole kernel: " + "  ".join(f"{k}={tot[k]}" for k in order if tot[k]))
        # one line per top-level loop: the widest extent of each header, loops nested in a reported one dropped
        widest = {}
        for tgt, a, b in lps:
            if tgt not in widest or b > widest[tgt][2]:
                widest[tgt] = (tgt, a, b)
        tops = []
        for tgt, a, b in sorted(widest.values(), key=lambda t: (t[1], -t[2])):
            if not any(a >= ta and b <= tb for _, ta, tb in tops):
                tops.append((tgt, a, b))
        for tgt, a, b in tops:
            c = collections.Counter(classify(mn, t) for mn, t in ins[a:b + 1])
            if c["fp64"] < args.min_fp64:
                continue
            valu = c["fp64"] + c["dpp"] + c["lane"] + c["cndmask"] + c["vcmp"] + c["valu_other"]
            print(f"   loop {tgt:<12} {b - a + 1:6d} insts  VALU={valu:5d}: " + "  ".join(f"{k}={c[k]}" for k in order if c[k]))


if __name__ == "__main__":
    sys.exit(main())
