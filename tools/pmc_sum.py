import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, "n=%d avg=%.0f" % (len(v), sum(v) / len(v)))
