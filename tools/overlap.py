"""Which kernels of the closed loop run at the same time: reads a rocprofv3 --kernel-trace directory and prints,
for the steady-state part of the trace, how long embedding and condense+QP launches ran alone and together."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
def kind(n):
    if "embed" in n: return "embed"
    if "mpc_solve" in n: return "solve"
    if "plant" in n: return "plant"
    return None
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"])) for r in rows if kind(r["Kernel_Name"]))
ev = ev[len(ev) * 4 // 10:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = sorted([(s, 1, k) for s, e, k in ev] + [(e, -1, k) for s, e, k in ev])
cur = {"embed": 0, "solve": 0, "plant": 0}; last = t0; hist = {}
for t, d, k in pts:
    key = tuple(sorted(n for n, c in cur.items() if c > 0))
    hist[key] = hist.get(key, 0) + (t - last); last = t; cur[k] += d
tot = t1 - t0
print("window %.2f ms, %d launches" % (tot / 1e6, len(ev)))
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]): print("  running %-22s %5.1f %%" % ("+".join(k) or "nothing", 100 * v / tot))
for name in ("embed", "solve", "plant"):
    d = np.array([(e - s) / 1e3 for s, e, n in ev if n == name])
    if len(d): print("  %s: n=%d mean %.1f us  max %.1f us  total %.2f ms (%.0f %% of the window)" % (name, len(d), d.mean(), d.max(), d.sum() / 1e3, 100 * d.sum() / tot))
