import csv, glob, sys, numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps end with plant_kernel
steps = []; cur = []
for s in seq:
    if "embed_kernel" in s[0] and "true" not in s[0].split("(")[0]: continue
    if not any(k in s[0] for k in ("embed_kernel", "mpc_solve_kernel", "plant_kernel")): continue
    cur.append(s)
    if "plant_kernel" in s[0]:
        steps.append(cur); cur = []
steps = [s for s in steps if len(s) == 9][-150:]
print("steps analysed", len(steps))
for i in range(9):
    d = [s[i][1] for s in steps]
    print(i, steps[0][i][0].split("(")[0][-40:], "mean %.1f us  p50 %.1f  max %.1f" % (np.mean(d), np.median(d), max(d)))
gaps = [np.sum([(s[j + 1][2] - s[j][3]) / 1e3 for j in range(8)]) for s in steps]
span = [(s[8][3] - s[0][2]) / 1e3 for s in steps]
inter = [(steps[i + 1][0][2] - steps[i][8][3]) / 1e3 for i in range(len(steps) - 1) ]
print("mean in-step gaps total %.1f us; step span %.1f us; between steps %.1f us" % (np.mean(gaps), np.mean(span), np.median(inter)))
