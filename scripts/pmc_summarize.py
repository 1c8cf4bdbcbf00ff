import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(path)):
    k = row.get("Kernel_Name", "")[:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print("   %-28s total %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
