"""Dev tool: kernel-boundary bubbles of one training step from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]
Prints, for the busiest queue, the busy time, the summed gaps between consecutive kernels and a histogram
of the gaps; then the union busy time over all queues."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
t0, t1 = ks[0][0], ks[-1][1]
lo = t0 + (t1 - t0) * skip
ks = [k for k in ks if k[0] >= lo]
span = ks[-1][1] - ks[0][0]
print(f"{len(ks)} kernels over {span/1e6:.3f} ms")
byq = collections.defaultdict(list)
for k in ks: byq[k[3]].append(k)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 20000]
    print(f"queue {q}: {len(lst)} kernels busy {busy/1e6:.3f} ms, gaps<20us: n={len(small)} sum {sum(small)/1e6:.3f} ms "
          f"median {sorted(small)[len(small)//2] if small else 0} ns; gaps>=20us: n={len(pos)-len(small)} sum {(sum(pos)-sum(small))/1e6:.3f} ms")
# union busy
ev = sorted([(s, 1) for s, e, _, _ in ks] + [(e, -1) for s, e, _, _ in ks])
depth = 0; last = None; busy = 0
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print(f"union busy {busy/1e6:.3f} ms of {span/1e6:.3f} ms ({100*busy/span:.1f}%)")
# gap after each kernel name (on its queue), aggregated
agg = collections.defaultdict(lambda: [0, 0])
for q, lst in byq.items():
    for i in range(len(lst) - 1):
        g = lst[i + 1][0] - lst[i][1]
        if 0 < g < 20000:
            n = lst[i][2].split("(")[0][:50]
            agg[n][0] += 1; agg[n][1] += g
for n, (c, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  after {n:50s} n={c:5d} gap sum {g/1e3:9.1f} us  avg {g/c:7.0f} ns")
