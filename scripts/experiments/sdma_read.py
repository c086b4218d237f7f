import csv,sys,glob
f=glob.glob('/tmp/sp/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
for r in rows:
    n=r['Kernel_Name']
    if 'k_marker' in n: print('marker', int(r['Grid_Size_X'])//64)
    else: print('    ', n[:40], r['Grid_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
