#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch.
usage: pmc_summary.py [--json out.json] <kernel name filter> <rocprofv3 output dir> ...
--json also writes {counter: {"mean": .., "n": ..}, "duration_us": {...}} over all matching dispatches."""
import csv, glob, sys, collections, json
def main(dirs, filt, jpath=None):
    tot = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                if filt in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                print(d, k)
                for c, v in sorted(cs.items()):
                    print("   %-32s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
        for f in glob.glob(d + "/*/*kernel_trace.csv"):
            t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if filt in r["Kernel_Name"]]
            if t:
                print("   %-32s n=%d mean=%.1f us" % ("duration", len(t), sum(t) / len(t) / 1e3))
                tot["duration_us"].extend(x / 1e3 for x in t)
    if jpath:
        json.dump({"kernel_filter": filt, **{c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(tot.items())}},
                  open(jpath, "w"), indent=1)
if __name__ == "__main__":
    a = sys.argv[1:]
    jp = None
    if a and a[0] == "--json":
        jp = a[1]; a = a[2:]
    main(a[1:], a[0], jp)
