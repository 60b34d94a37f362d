"""Dev tool: the kernels of one step between two name patterns, in start order, with durations and gaps
(rocprofv3 --kernel-trace csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
a, b = sys.argv[2], sys.argv[3]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if "adam_flat" in e[2]]
lo, hi = adam[-2], adam[-1]
step = ev[lo + 1:hi + 1]
i0 = next(i for i, e in enumerate(step) if a in e[2])
i1 = max(i for i, e in enumerate(step) if b in e[2])
prev = step[i0][0]
for s, e, n, q in step[i0:i1 + 1]:
    print(f"{(s - step[i0][0]) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  q{q:>3s}  {n.split('(')[0][-70:]}")
    prev = max(prev, e)
print(f"span {(step[i1][1] - step[i0][0]) / 1e3:.1f} us, {i1 - i0 + 1} kernels")
