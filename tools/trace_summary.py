"""Summarise a rocprofv3 kernel-trace CSV per (kernel, grid, workgroup) -- separates the shapes of one kernel."""
import csv, sys, collections, glob

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-48:]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
               r.get("LDS_Block_Size", "?"))
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':50s} {'grid':>8s} {'wg':>5s} {'lds':>7s} {'calls':>7s} {'avg_us':>8s} {'min_us':>8s} {'p50_us':>8s} {'share':>6s}")
for (name, grid, wg, lds), v in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    v.sort()
    print(f"{name:50s} {grid:>8s} {wg:>5s} {lds:>7s} {len(v):7d} {sum(v)/len(v)/1e3:8.2f} {v[0]/1e3:8.2f} {v[len(v)//2]/1e3:8.2f} {sum(v)/tot:6.3f}")
