"""Per-round kernel time table of the last bench step from a rocprofv3 kernel trace (csv)."""
import csv, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(d + '_kernel_stats.csv')))
for r in rows[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
rows = list(csv.DictReader(open(d + '_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'asm_count_k' in n]
seg = rows[idx[-34]:]
rnd = -1; acc = {}
for r in seg:
    n = r['Kernel_Name']; t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'asm_count_k' in n:
        rnd += 1; acc[rnd] = {'t0': int(r['Start_Timestamp'])}
    key = ('reg' if 'reg_k' in n else 'tile0' if 'tile_k<0>' in n else 'tile1' if 'tile_k<1>' in n else 'gemm' if 'gemm' in n
           else 'upd' if 'update' in n else 'count' if 'count' in n else 'other')
    acc[rnd].setdefault(key, 0); acc[rnd][key] += t; acc[rnd]['t1'] = int(r['End_Timestamp'])
for k in sorted(acc)[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    a = acc[k]; print(k, 'wall', round((a['t1'] - a['t0']) / 1e3), {x: round(y) for x, y in a.items() if x not in ('t0', 't1')})
