"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic of the ICP pass kernel.
MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
(16 B/lane) reads -> doubled here; WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import sys


def mean_counter(d, counter, kernel_substr):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return vals


def main(fetch_dir, write_dir, out):
    res = {}
    for k in ("icp_fused_kernel", "icp_accumulate_kernel", "icp_reduce_update_kernel"):
        f = mean_counter(fetch_dir, "FETCH_SIZE", k)
        w = mean_counter(write_dir, "WRITE_SIZE", k)
        if k == "icp_fused_kernel":  # drop the one-workgroup tail-only launches that close a registration (no pass in them)
            f = [v for v in f if v > 64.0] or f
        if not f or not w:
            continue
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        res[k] = {"launches": len(f), "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm,
                  "hbm_bytes_per_launch_corrected": (2.0 * fm + wm) * 1024.0,
                  "note": "FETCH_SIZE doubled (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section); mean over all passes of the "
                          "bench (1 misaligned first pass + 10 converged passes per step)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
