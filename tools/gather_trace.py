"""Durations (us) of the gather_segments launches of the last captured step in a rocprofv3 kernel trace:
   python tools/gather_trace.py <rocprof output dir>"""
import csv,re,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[re.sub(r'\(.*','',r['Kernel_Name'])[:48] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('sgd_flat')]
a,b=idx[-2],idx[-1]
seg=rows[a+1:b+1]
g=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in seg if 'gather_segments' in r['Kernel_Name']]
print(len(g),[round(x,1) for x in g], round(sum(g),1))
