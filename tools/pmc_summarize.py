import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else 'k_apply'
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            key = (r['Kernel_Name'].split('(')[0][-24:], r['Counter_Name'])
            acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
for k, (v, n) in sorted(acc.items()):
    print(k[0], k[1], 'avg/dispatch', v / n, 'n', n)
