#!/usr/bin/env python
"""profiles/r06_pmc_traffic.json (or the path given as argument) from the per-dispatch counter rows of scripts/gpu_pmc_traffic.sh (gpurun_out/pmc_traffic/m1_{dram,wr}):
what bench.py quotes as roofline.traffic.  The calibration behind the byte-per-request figures is profiles/r03_pmc_calibration*.txt."""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"source": "rocprofv3 --kernel-trace --pmc, one counter set per run, `python bench.py --steps 20 --warmup 3 --m2-frames 0 --no-cpu-baseline --concurrent 0 --no-gicp --no-host-seam` "
                 "(scripts/gpu_pmc_traffic.sh); per-launch means over the pass launches (the one-workgroup tail launches of the fused loop dropped)",
       "calibration": {"file": "profiles/r03_pmc_calibration_and_m1_traffic.txt",
                       "read_request_bytes": "TCC_EA0_RDREQ counts ONE request per 128-B line for coalesced / fully used lines (stream, aligned 128-B runs: requested bytes / "
                                             "requests = 128.0 / 128.4) and one per touched 64-B half line for sparse access (aligned 64-B runs: 64.2; single 16-B records: 16.06 "
                                             "requested bytes per request); FETCH_SIZE = requests x 64 B, i.e. half the bytes of a streaming read (the guide's x2) and the "
                                             "bytes of a sparse one",
                       "write_request_bytes": "TCC_EA0_WRREQ = TCC_EA0_WRREQ_64B: 64.0 requested bytes per request for streaming 16-B stores",
                       "infinity_cache": "cold and warm dispatches give identical counts at 32 MiB, 128 MiB and 1 GiB working sets, and TCC_EA0_RDREQ_DRAM equals TCC_EA0_RDREQ "
                                         "everywhere: the counters sit at the L2 <-> fabric boundary, Infinity-Cache hits are included and cannot be separated from HBM reads"},
       "kernels": {}}
def src_hash(*names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(ROOT, "open3d_slam_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]
# the figures belong to the kernel sources they were measured on: bench.py quotes them only while the hash still matches
out["kernel_source_sha16"] = {"icp_kernels.hpp": src_hash("icp_kernels.hpp"), "stream (cloud_kernels.hpp + normals_kernel.hpp + map_kernels.hpp)": src_hash("cloud_kernels.hpp", "normals_kernel.hpp", "map_kernels.hpp")}
def rows(setname, pat, run="m1"):
    f = glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_traffic/{run}_{setname}/**/*counter_collection.csv"), recursive=True)[0]
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {c: [x[1] for x in sorted(v)] for c, v in by.items()}
def stat(a):
    m = sorted(a)[len(a) // 2]
    keep = [x for x in a if x >= 0.25 * m] if m > 0 else a
    return {"launches": len(keep), "mean": sum(keep) / max(len(keep), 1), "median": m, "max": max(a)}
ALGO = 65536 * 228


def kernel_key(name):
    """`void o3ds::(anonymous namespace)::foo_kernel<o3ds::P4f>(args)` -> `foo_kernel<P4f>`: cut at the ARGUMENT list's parenthesis, not at
    the one of `(anonymous namespace)` (round 5 filed every such kernel under the key "")"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("o3ds::", "")
    return name.split("(")[0][:60]


# launches of the legs of `bench.py --steps 20 --warmup 3` in the order they run (12 per registration: 11 passes + the fold launch): a leg is
# warm-up + timed steps + the bracketed re-run + the span re-run (round 6) registrations
STEPS, WARMUP = 20, 3
N32 = (WARMUP + STEPS + 2 * min(STEPS, 100)) * 12
SB = max(STEPS // 4, 5)
NBIG = (3 + SB + 2 * min(SB, 100)) * 12  # (the in-library one-rank leg that follows uses the same instantiation: cut off)
for name, pat, sl in (("icp_fused_kernel<P4f> configs[1] (1 M-point map)", "icp_fused_kernel<o3ds::P4f", slice(0, N32)),
                      ("icp_fused_kernel<P4f> 8 M-point map", "icp_fused_kernel<o3ds::P4f", slice(N32, N32 + NBIG)),
                      ("icp_fused_kernel<P4d> configs[1] (f64 storage)", "icp_fused_kernel<o3ds::P4d", slice(None))):
    rd, wr = stat(rows("dram", pat)["TCC_EA0_RDREQ_sum"][sl]), stat(rows("wr", pat)["TCC_EA0_WRREQ_sum"][sl])
    lo = rd["mean"] * 64 + wr["mean"] * 64
    hi = rd["mean"] * 128 + wr["mean"] * 64
    out["kernels"][name] = {"read_requests_per_launch": rd, "write_requests_per_launch": wr,
                            "traffic_bytes_per_launch": lo, "traffic_bytes_per_launch_if_every_read_is_a_full_line": hi,
                            "algorithmic_bytes_per_launch": ALGO, "traffic_over_algorithmic": [lo / ALGO, hi / ALGO],
                            "compulsory_bytes_per_launch": 65536 * 48, "traffic_over_compulsory": [lo / (65536 * 48), hi / (65536 * 48)]}
# ---- the kernels of a configs[2] frame (scripts/stream_few_frames.py under the same two counter sets): per-launch means by kernel
out["stream_kernels"] = {}
try:
    def all_rows(setname):
        f = glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_traffic/stream_{setname}/**/*counter_collection.csv"), recursive=True)[0]
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = kernel_key(r["Kernel_Name"])
            by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return by
    rd_by, wr_by = all_rows("dram"), all_rows("wr")
    for k in sorted(rd_by):
        if "rocprim" in k or "rocclr" in k or k not in wr_by:
            continue
        rd, wr = rd_by[k].get("TCC_EA0_RDREQ_sum", []), wr_by[k].get("TCC_EA0_WRREQ_sum", [])
        if not rd or not wr:
            continue
        mr, mw = sum(rd) / len(rd), sum(wr) / len(wr)
        out["stream_kernels"][k] = {"launches": len(rd), "read_requests_per_launch": mr, "write_requests_per_launch": mw,
                                    "traffic_bytes_per_launch": mr * 64 + mw * 64, "traffic_bytes_per_launch_if_every_read_is_a_full_line": mr * 128 + mw * 64}
except Exception as e:  # the stream sets are optional
    out["stream_kernels_error"] = repr(e)
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join("profiles", "r06_pmc_traffic.json")
json.dump(out, open(os.path.join(ROOT, dst), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
