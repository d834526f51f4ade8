"""Prints rocprofv3's kernel_stats.csv with short kernel names: python3 scripts/kstats.py <dir> [rows]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for i, r in enumerate(csv.DictReader(open(path))):
    if i >= rows: break
    name = r["Name"].replace("void ", "")
    print(f'{name[:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  {float(r["Percentage"]):5.1f} %')
