"""Per-launch timeline of the last solve of a rocprofv3 kernel trace: python scripts/timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if 'pad_x0_k' in r['Kernel_Name']]
start = idx[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0 = int(rows[start]['Start_Timestamp'])
tot = {}
for r in rows[start:]:
    n = r['Kernel_Name'].split('(')[0].replace('nnmpc::', '').replace('void ', '').replace('(anonymous namespace)::', '')
    if 'pad_x0_k' in n and r is not rows[start]:
        break
    s = (int(r['Start_Timestamp']) - t0) / 1e6
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    tot[n] = tot.get(n, 0.0) + d
    print(f"{s:8.3f} {d:7.3f} {n[:44]:44s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])} q{r['Queue_Id']}")
print("---- per kernel (ms)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v:8.3f} {k}")
