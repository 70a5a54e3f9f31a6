"""Per repetition of the odometry step in a kernel trace: when the two k_knn_tree start / end relative to the first k_features, and
when the build's k_pair_keys starts.  python profiles/dev/step_reps.py <b_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = []
for r in rows:
    n = r['Kernel_Name']
    g = int(r['Grid_Size_X'])
    s, e = int(r['Start_Timestamp']) / 1000, int(r['End_Timestamp']) / 1000
    if 'k_features' in n and g == 250112: ev.append(('feat', s, e, r['Queue_Id']))
    elif 'k_knn_tree' in n and g == 249984: ev.append(('knn', s, e, r['Queue_Id']))
    elif 'k_pair_keys' in n and g in (249600, 249344): ev.append(('keys', s, e, r['Queue_Id']))
    elif 'k_sorted_feat' in n and g == 62720: ev.append(('sfeat62k', s, e, r['Queue_Id']))
i = 0
while i < len(ev):
    if ev[i][0] != 'feat': i += 1; continue
    t0 = ev[i][1]; j = i + 1; knn = []; keys = None; sf = None
    while j < len(ev) and ev[j][0] != 'feat':
        if ev[j][0] == 'knn': knn.append(ev[j])
        if ev[j][0] == 'keys' and keys is None: keys = ev[j]
        if ev[j][0] == 'sfeat62k': sf = ev[j]
        j += 1
    if len(knn) == 2 and keys:
        print("step: " + "  ".join("knn q%s %6.0f -> %6.0f (%5.0f)" % (k[3], k[1] - t0, k[2] - t0, k[2] - k[1]) for k in knn)
              + ("  sorted_feat(fix) %5.0f us" % (sf[2] - sf[1]) if sf else "") + "  build starts %6.0f" % (keys[1] - t0))
    i = j
