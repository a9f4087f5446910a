"""Phase profile of one steady icp_fused_kernel launch from the O3DS_FUSED_TRACE dump (100 MHz wall clock, thread 0 of each workgroup).
stamps: 0 start | 1 state+slots loaded | 2 step done | 3 pass body done | 4,5 record added to its slot | 6 end ; col 7 = last arriver (older epilogue)"""
import sys
import numpy as np

t = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0 = t[:, 0].min()
us = (t[:, :7] - t0) / 100.0
names = ["start", "loaded", "stepped", "body", "published", "ticket", "end"]
print("workgroups", len(t), " last arrivers", int(t[:, 7].sum()))
print("absolute time since first workgroup start (us): min / mean / max")
for k, n in enumerate(names):
    print("  %-10s %6.2f %6.2f %6.2f" % (n, us[:, k].min(), us[:, k].mean(), us[:, k].max()))
d = np.diff(us, axis=1)
print("phase durations (us): mean / p95 / max")
for k in range(6):
    print("  %-22s %6.2f %6.2f %6.2f" % (names[k] + "->" + names[k + 1], d[:, k].mean(), np.percentile(d[:, k], 95), d[:, k].max()))
la = t[:, 7] == 1
if la.any():  # only the ticket-and-fold epilogue had last arrivers; the exact-atomics epilogue has none
    print("last arrivers: ticket->end mean %.2f max %.2f ; others %.2f" % (d[la, 5].mean(), d[la, 5].max(), d[~la, 5].mean()))
st = (t[:, [8, 10, 11, 12]] - t[:, 1:2]) / 100.0
print("inside the step, us after 'loaded' (mean): enter %.2f | solved %.2f | sincos %.2f | done %.2f" % tuple(st.mean(0)))
bd = (t[:, 13:16] - t[:, 2:3]) / 100.0
print("inside the body, us after 'stepped' (mean / max): group search %.2f / %.2f | far loop %.2f / %.2f | records+barrier %.2f / %.2f | body end %.2f / %.2f" % (
    bd[:, 0].mean(), bd[:, 0].max(), bd[:, 1].mean(), bd[:, 1].max(), bd[:, 2].mean(), bd[:, 2].max(), d[:, 2].mean(), d[:, 2].max()))
