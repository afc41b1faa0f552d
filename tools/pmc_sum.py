"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python tools/pmc_sum.py <dir> (dev helper)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = len(launches[(k, c)]); print("   %-28s %14.1f M total  %12.3f M / launch (%d launches)" % (c, acc[k][c] / 1e6, acc[k][c] / 1e6 / n, n))
