"""Sum rocprofv3 --pmc counter_collection.csv per kernel (last dispatch):  python tools/pmc_parse.py dir [substr]"""
import csv, collections, glob, sys
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    last = {}
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = r["Kernel_Name"][:60]
            agg[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            last[k] = r["Dispatch_Id"]
    for k, d in last.items():
        print(k)
        for c, v in sorted(agg[(k, d)].items()):
            print(f"   {c:32s} {v:16.0f}")
