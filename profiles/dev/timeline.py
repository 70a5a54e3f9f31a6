"""Timeline of the tail of a rocprofv3 kernel trace: python profiles/dev/timeline.py <b_kernel_trace.csv> [window_us] [min_us]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 6200.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::', 'rp::', n)
    n = re.sub(r'\(.*', '', n)
    return n[:60]
t_end = int(rows[-1]['End_Timestamp'])
sel = [r for r in rows if int(r['Start_Timestamp']) > t_end - win * 1000]
t0 = int(sel[0]['Start_Timestamp'])
last = None
for r in sel:
    s = (int(r['Start_Timestamp']) - t0) / 1000; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
    key = (nm(r), r['Queue_Id'])
    if key == last and d < min_us:
        continue
    last = key
    if d >= min_us:
        print(f"{s:9.1f} +{d:8.1f}  q{r['Queue_Id']}  grid {r.get('Grid_Size_X','?'):>9} wg {r.get('Workgroup_Size_X','?'):>4}  {key[0]}")
