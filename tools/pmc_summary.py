"""Summarise a tools/profile.sh output directory: per-kernel time stats and PMC counter means."""
import csv, glob, os, sys, json, collections

def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r

def main(out):
    res = {"kernels": {}, "counters": {}}
    for f, r in rows(os.path.join(out, "stats", "**", "*kernel_stats.csv")):
        res["kernels"][r["Name"][:120]] = {k: r[k] for k in r if k != "Name"}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f, r in rows(os.path.join(out, "pmc_*", "**", "*counter_collection.csv")):
        acc[r["Kernel_Name"][:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        res["counters"][k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    with open(os.path.join(out, "summary.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in res["kernels"].items():
        print("K", k[:90], v)
    for k, v in res["counters"].items():
        print("C", k[:90], {c: round(x["mean"], 1) for c, x in v.items()})

main(sys.argv[1])
