import csv, glob, sys, numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
      if any(k in r["Kernel_Name"] for k in ("embed_kernel<2, 1, 1, true", "mpc_solve_kernel", "plant_kernel"))]
ev.sort()
# take the last 60% of events (steady state)
ev = ev[len(ev) * 4 // 10:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
# union busy time and concurrency histogram
pts = sorted([(s, 1) for s, e, n in ev] + [(e, -1) for s, e, n in ev])
cur = 0; last = t0; hist = {}
for t, d in pts:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
tot = t1 - t0
print("window %.1f ms, kernels %d" % (tot / 1e6, len(ev)))
for k in sorted(hist): print("  %d kernels running: %.1f %%" % (k, 100 * hist[k] / tot))
for name in ("embed", "solve", "plant"):
    d = [(e - s) / 1e3 for s, e, n in ev if name in n]
    print("  %s: n=%d mean %.1f us total %.1f ms" % (name, len(d), np.mean(d), sum(d) / 1e3))
