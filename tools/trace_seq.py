import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)//4
last=rows[-n:]
tot=0
for r in last:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=d
    print(f"{d:7.1f} {r['Kernel_Name'][:110]} g={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print('launches',n,'sum us',tot)
