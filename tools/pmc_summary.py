#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
def main(dirs, filt):
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                if filt in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                print(d, k)
                for c, v in sorted(cs.items()):
                    print("   %-32s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
        for f in glob.glob(d + "/*/*kernel_trace.csv"):
            t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if filt in r["Kernel_Name"]]
            if t: print("   %-32s n=%d mean=%.1f us" % ("duration", len(t), sum(t) / len(t) / 1e3))
if __name__ == "__main__":
    main(sys.argv[2:], sys.argv[1])
