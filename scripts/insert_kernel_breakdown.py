"""Per-kernel time of o3ds_map_insert_scan early and late in the growth of a map, from a rocprofv3 --kernel-trace --output-format csv
directory of `python scripts/insert_sweep_run.py`: insertions are told apart by their pm_place_kernel; prints the mean duration of every
kernel of an insertion for an early and a late window."""
import collections, csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
marks = [k for k, r in enumerate(rows) if "pm_place_kernel" in r[0]]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"o3ds::", "", n); return n.split("<")[0].split("(")[0]
def window(a, b):
    acc = collections.defaultdict(float); cnt = 0
    for i in range(a, min(b, len(marks) - 1)):
        lo, hi = marks[i], marks[i + 1]
        for name, s, e in rows[lo:hi]:
            sn = short(name)
            if sn.startswith("pm_") or sn == "vox_order_kernel":
                acc[sn] += (e - s) / 1e3
        cnt += 1
    return {k: v / max(cnt, 1) for k, v in acc.items()}, cnt
for (a, b) in ((5, 35), (len(marks) - 40, len(marks) - 1)):
    w, c = window(a, b)
    print("insertions %d..%d (%d): total %.1f us  " % (a, b, c, sum(w.values())) + "  ".join("%s %.1f" % (k, v) for k, v in sorted(w.items())))
