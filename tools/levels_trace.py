import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*_kernel_trace.csv',recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'k_scatter_binned' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-16*20:]
import collections
acc=collections.defaultdict(list)
for i,r in enumerate(rows): acc[i%16].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for l in range(16): print(l, round(sum(acc[l])/len(acc[l]),1))
