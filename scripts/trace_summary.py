"""Per-iteration timeline from a rocprofv3 kernel_trace.csv: python3 scripts/trace_summary.py <csv> [last-iteration only]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split at the marker kernel (elementwise add on a 1-element tensor is not unique by name; use the LAST occurrences of the
# first mf:: kernel instead): take the last full iteration = from the last-but-one to the last 'add' preceding an mf kernel
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "mf::" in n]
if not idx:
    sys.exit("no mf:: kernels")
# iterations: find marker positions = elementwise kernels immediately followed (within 30 kernels) by first mf kernel; simpler: cut at gaps
starts = [i + 1 for i, n in enumerate(names) if "CUDAFunctorOnSelf_add<float>" in n]   # the marker of scripts/prof_cfg.py
if len(starts) >= 2:
    starts.append(len(rows))
# the marker is somewhere before each start; an iteration = [starts[k]-back .. starts[k+1]-back)
if len(starts) < 2:
    lo, hi = 0, len(rows)
else:
    lo, hi = starts[-2], starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
tot_busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    n = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f}  gap {gap:7.1f}  {n}")
    prev_end = e
    tot_busy += e - s
print(f"iteration span {(int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3:.1f} us, busy {tot_busy / 1e3:.1f} us, kernels {hi - lo}")
