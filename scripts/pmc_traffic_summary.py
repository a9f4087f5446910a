#!/usr/bin/env python
"""Summary of scripts/gpu_pmc_traffic.sh: the calibration table (counter readings against known byte counts) and the per-launch counter
readings of the backend's kernels.  Prints text; with --json also writes the numbers bench.py quotes as roofline.traffic."""
import collections
import csv
import glob
import json
import os
import re
import sys

out_dir = sys.argv[1]
want_json = len(sys.argv) > 2 and sys.argv[2] == "--json"


def rows(run):
    fs = glob.glob(os.path.join(out_dir, run, "**", "*counter_collection.csv"), recursive=True)
    for f in fs:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def per_kernel(run):
    """kernel name -> counter -> list of (dispatch id, value) in dispatch order"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(run):
        name = r["Kernel_Name"]
        agg[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k in agg:
        for c in agg[k]:
            agg[k][c].sort()
    return agg


print("== counters available on this box (rocprofv3 -L, filtered) ==")
try:
    print(open(os.path.join(out_dir, "counters_available.txt")).read()[:3000])
except OSError:
    pass
print("== calibration: plain run (no profiler) ==")
calib = {}
try:
    for ln in open(os.path.join(out_dir, "calib_plain.txt")):
        print(ln.rstrip())
        m = re.match(r"\s*(\d+) (\S+)\s+ws_MiB\s+(\d+) requested_MB\s+([\d.]+)", ln)
        if m:
            calib[int(m.group(1))] = dict(pattern=m.group(2), ws=int(m.group(3)), req=float(m.group(4)) * 1e6)
except OSError:
    pass

print("\n== calibration: counter readings per dispatch (cold, warm) against the requested bytes ==")
table = collections.defaultdict(dict)
for run in ("calib_fetch", "calib_write", "calib_ea", "calib_hit", "calib_dram", "calib_wr"):
    for k, cs in per_kernel(run).items():
        m = re.search(r"calib_kernel<(\d+)", k)
        if not m:
            continue
        cid = int(m.group(1))
        for c, v in cs.items():
            table[cid][c] = [x[1] for x in v]
counters = sorted({c for t in table.values() for c in t})
print("%3s %-8s %6s %12s | " % ("id", "pattern", "ws", "requested") + " | ".join("%-28s" % c for c in counters))
for cid in sorted(table):
    info = calib.get(cid, dict(pattern="?", ws=0, req=0.0))
    cells = []
    for c in counters:
        v = table[cid].get(c, [])
        cells.append("%-28s" % ("/".join("%.4g" % x for x in v[:2])))
    print("%3d %-8s %6d %12.4g | " % (cid, info["pattern"], info["ws"], info["req"]) + " | ".join(cells))
print("\nratios requested_bytes / counter (cold dispatch):")
for cid in sorted(table):
    info = calib.get(cid)
    if not info:
        continue
    parts = []
    for c in counters:
        v = table[cid].get(c, [])
        if v and v[0] > 0:
            parts.append("%s %.4g (warm %.4g)" % (c, info["req"] / v[0], info["req"] / v[1] if len(v) > 1 and v[1] > 0 else float("nan")))
    print("%3d %-8s ws %5d MiB: " % (cid, info["pattern"], info["ws"]) + "; ".join(parts))


def kernel_table(prefix, label, select):
    print(f"\n== {label}: mean counter reading per launch ==")
    res = {}
    for setname in ("fetch", "write", "ea", "dram", "wr"):
        for k, cs in per_kernel(f"{prefix}_{setname}").items():
            short = k.split("(")[0]
            if not any(s in short for s in select):
                continue
            for c, v in cs.items():
                vals = [x[1] for x in v]
                srt = sorted(vals)
                med = srt[len(srt) // 2]
                big = [x for x in vals if x >= 0.25 * med] if med > 0 else vals  # drop the one-workgroup tail launches of the fused loop
                res.setdefault(short[-70:], {})[c] = dict(n=len(vals), n_counted=len(big), mean=sum(big) / max(len(big), 1), median=med, max=srt[-1])
    for k in sorted(res):
        for c, d in sorted(res[k].items()):
            print("%-70s %-24s n=%5d (counted %5d) mean=%14.1f median=%14.1f max=%14.1f" % (k, c, d["n"], d["n_counted"], d["mean"], d["median"], d["max"]))
    return res


m1 = kernel_table("m1", "bench.py M1 (65 536 queries vs 1 M map)", ("icp_fused", "icp_accumulate"))
st = kernel_table("stream", "configs[2] stream, 40 frames", ("normals", "scatter", "merge_", "icp_fused", "vox_", "segment_", "cell_count", "crop"))
if want_json:
    json.dump(dict(calibration={str(k): dict(calib.get(k, {}), counters=v) for k, v in table.items()}, m1=m1, stream=st),
              open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
