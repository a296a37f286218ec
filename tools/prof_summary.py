"""Summarise a rocprofv3 --kernel-trace CSV: per-step time of each ggms kernel (diagnostic tool)."""
import collections, csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 23
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
names = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    names.setdefault(r['Kernel_Name'][:86], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    if 'ggms' in k or 'rocclr' in k:
        tot += sum(v) / steps
        print(f"{k:86s} n={len(v):4d} per_step_us={sum(v)/steps:8.1f} last={[round(x,1) for x in v[-4:]]}")
print("total per step us:", round(tot, 1))
