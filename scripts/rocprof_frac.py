#!/usr/bin/env python3
"""Recompute bench.py's roofline fraction from a rocprofv3 kernel trace of the same command.

    python scripts/rocprof_frac.py <..._kernel_trace.csv> [--gbytes G] [--out f.json]

A forward is the run of dispatches from one launch of the network's first kernel (y5_conv_front_kernel where the fused front is the plan's
choice, else y5_conv_stem_kernel) to the next; the conv launches inside it (front / igemm / h3 / pw / k3 / stem / bneck / pw_head) are summed per forward from the trace's own begin/end timestamps (pure kernel
durations: no dispatch gaps, so this sum is a lower bound of the in-situ event-to-event figure bench.py uses).  Groups that
are not whole forwards (autotune bursts, isolated per-op timing) have a different launch count and are dropped by keeping the
most common count only.  mfma_frac = algorithmic GFLOP / median sum / 2500 TFLOP/s (bench.py roofline.frac);
hbm frac = algorithmic GB (bench.py's `algorithmic_gbytes_per_step`) / median sum / 8000 GB/s.
"""
import argparse, collections, csv, json, statistics, sys

CONV = ("conv_igemm", "conv_g8", "conv_h3", "conv_pw", "conv_k3", "conv_pwk", "conv_stem", "conv_bneck", "conv_front", "sppf_cv1_pool", "conv_headk")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--gbytes", type=float, default=None, help="algorithmic GB per forward (bench.py roofline.algorithmic_gbytes_per_step)")
    ap.add_argument("--gflop", type=float, default=None, help="algorithmic GFLOP per forward (bench.py roofline.algorithmic_gflop_per_step): the MFMA fraction")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        k = r["Kernel_Name"]
        if "y5_" not in k:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    # the kernel that opens a forward: the fused front if the trace has MORE launches of it than of the stand-alone stem (whose remaining launches
    # are then the plan builder's timing of the alternative), else the stem
    n_front = sum("conv_front" in k for _, _, k in rows)
    n_stem = sum("conv_stem" in k for _, _, k in rows)
    first = "conv_front" if n_front > n_stem else "conv_stem"
    groups, cur = [], None
    for s, e, k in rows:
        if first in k:
            if cur:
                groups.append(cur)
            cur = []
        if cur is not None and any(c in k for c in CONV):
            cur.append((k, e - s, s, e))
    if cur:
        groups.append(cur)
    cnt = collections.Counter(len(g) for g in groups)
    n, _ = cnt.most_common(1)[0]
    fw = [g for g in groups if len(g) == n]
    sums = [sum(d for _, d, _, _ in g) / 1e6 for g in fw]                         # ms of kernel time per forward
    spans = [(g[-1][3] - g[0][2]) / 1e6 for g in fw]                              # first conv start -> last conv end
    per = collections.defaultdict(list)
    for g in fw:
        for k, d, _, _ in g:
            per[k].append(d)
    out = {"forwards": len(fw), "conv_launches_per_forward": n,
           "conv_kernel_ms_per_forward": {"median": round(statistics.median(sums), 4), "min": round(min(sums), 4), "max": round(max(sums), 4)},
           "first_to_last_conv_ms": round(statistics.median(spans), 4),
           "kernels": sorted(({"kernel": k[:110], "launches_per_forward": round(len(v) / len(fw), 2), "avg_us": round(sum(v) / len(v) / 1e3, 2)}
                              for k, v in per.items()), key=lambda r: -r["avg_us"] * r["launches_per_forward"])}
    if a.gbytes:
        out["algorithmic_gbytes_per_forward"] = a.gbytes
        out["frac_from_trace"] = round(a.gbytes / statistics.median(sums) * 1e3 / 8000.0, 4)
    if a.gflop:
        out["algorithmic_gflop_per_forward"] = a.gflop
        out["mfma_tflops_from_trace"] = round(a.gflop / statistics.median(sums), 2)           # GFLOP / ms = TFLOP/s
        out["mfma_frac_from_trace"] = round(a.gflop / statistics.median(sums) / 2500.0, 4)   # bench.py roofline.frac (round 3 definition)
    s = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(s)
    print(s)


if __name__ == "__main__":
    sys.exit(main())
