#!/bin/bash
# temporary experiment: Schur kernel phase ablation
for e in 0 1 2 4 8 16 6 7 31; do
  export TB_BA_EXP=$e
  tools/rocprof_stats.sh pba$e tools/prof_ba.py 64 1 > /dev/null
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/prof_pba$e/pba${e}_kernel_trace.csv')) if 'schur' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('exp=$e', 'n=%d'%len(d), 'first3', [round(x,1) for x in d[:3]])
PY
done
