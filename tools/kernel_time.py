import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*_kernel_stats.csv',recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in sys.argv[2:]): print(sys.argv[1].split('/')[-1], r['Name'][:40], round(float(r['AverageNs'])/1e3,1))
