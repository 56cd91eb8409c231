import csv, sys, glob, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-40s n=%3d mean=%.4g' % (c, len(v), sum(v) / len(v)))
