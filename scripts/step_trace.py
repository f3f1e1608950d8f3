"""Kernel timeline of the last bench step from a rocprofv3 kernel trace (csv): per-kernel totals of that step."""
import csv, sys, collections
d = sys.argv[1]
rows = list(csv.DictReader(open(d + '_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
cert = [i for i, n in enumerate(names) if 'asm_certify_k' in n]
seg = rows[cert[-2] + 1:cert[-1] + 1]
wall = (int(seg[-1]['End_Timestamp']) - int(rows[cert[-2]]['End_Timestamp'])) / 1e3
tot = collections.Counter(); calls = collections.Counter()
for r in seg:
    n = r['Kernel_Name'].replace('nnmpc::', '').split('(')[0][:40]
    tot[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; calls[n] += 1
print("step wall us", round(wall))
for n, t in tot.most_common(): print(f"{n:42s} {calls[n]:4d} {t:9.0f} us {100 * t / wall:5.1f} %")
if len(sys.argv) > 2:
    t0 = int(seg[0]['Start_Timestamp'])
    for r in seg: print(round((int(r['Start_Timestamp']) - t0) / 1e3), round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3), r['Kernel_Name'].replace('nnmpc::', '')[:30], r['Grid_Size_X'], r['Grid_Size_Y'])
