"""One lidar frame of the stream as the GPU saw it: every dispatch between two consecutive frame ingests (pack_strided_f32_kernel), with its
start (us after the frame's first dispatch), duration and the idle gap in front of it.  From a rocprofv3 --kernel-trace rocpd database."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"o3ds::", "", n)
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    return n.split("(")[0][:70]


def main(db_path, out_path, frame=60):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name,start,end from kernels order by start"))
    # a frame = from the first kernel of a scan's pre-processing (vox_insert_kernel: one per scan, the odometry and the mapper share the
    # pre-processed cloud) to the next one; the ingest of the NEXT scan (copy stream: pack_strided_f32_box_kernel, box_publish_kernel)
    # runs inside it, beside the frame's own kernels
    marks = [k for k, r in enumerate(rows) if "vox_insert_kernel" in r[0]]
    a, b = marks[frame], marks[frame + 1]
    t0 = rows[a][1]
    lines = ["# dispatches of stream frame %d (from its first pre-processing kernel to the next frame's): start_us dur_us gap_us kernel" % frame]
    prev_end = None
    busy = 0.0
    for name, s, e in rows[a:b]:
        gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
        lines.append("%9.1f %8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(name)))
        busy += (e - max(s, prev_end or s)) / 1e3 if (prev_end is None or e > prev_end) else 0.0  # (union of the intervals: two streams)
        prev_end = e if prev_end is None else max(prev_end, e)
    span = (rows[b][1] - t0) / 1e3
    lines.append("# %d dispatches, %.1f us busy of %.1f us between the two frame starts (%.0f %%)" % (b - a, busy, span, 100.0 * busy / span))
    # ... and the same two figures over many frames (a single frame may be a carving frame or not)
    tot_busy = tot_span = 0.0
    lo_f, hi_f = 20, min(len(marks) - 2, 90)
    for f in range(lo_f, hi_f):
        prev_end, fb = None, 0.0
        for name, s, e in rows[marks[f]:marks[f + 1]]:
            if prev_end is None or e > prev_end:
                fb += (e - max(s, prev_end or s)) / 1e3
            prev_end = e if prev_end is None else max(prev_end, e)
        tot_busy += fb
        tot_span += (rows[marks[f + 1]][1] - rows[marks[f]][1]) / 1e3
    if hi_f > lo_f:
        lines.append("# frames %d..%d: %.1f us busy of %.1f us per frame on average (%.0f %%)" % (lo_f, hi_f - 1, tot_busy / (hi_f - lo_f), tot_span / (hi_f - lo_f),
                                                                                                  100.0 * tot_busy / tot_span))
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
