import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, a in agg.items():
    if 'k_segments' not in k and 'k_trace' not in k: continue
    print(k)
    for c, v in sorted(a.items()): print("   ", c, f"{v:.4g}")
    if a.get('SQ_ACTIVE_INST_VALU'):
        print("    lane util", a['SQ_THREAD_CYCLES_VALU'] / (a['SQ_ACTIVE_INST_VALU'] * 64))
    if a.get('SQ_WAVE_CYCLES'):
        for c in ('SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS'):
            if c in a: print("    ", c, "/ wave_cycles", a[c] / a['SQ_WAVE_CYCLES'])
