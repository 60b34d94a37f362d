"""Dev tool: one steady-state training step from a rocprofv3 --kernel-trace CSV as a timeline: per queue busy time,
time with 1 / 2 / 3+ kernels resident, and (optionally) the kernels of a time window with their queues.
usage: python tools/trace_timeline.py <kernel_trace.csv> [t_from_ms t_to_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void avid::", "").replace("avid::", "")[:44], r.get("Queue_Id", "0")) for r in rows))
# steps are delimited by adam_flat_kernel
ad = [i for i, k in enumerate(ks) if k[2].startswith("adam_flat")]
if len(ad) < 3: sys.exit("need >= 3 steps")
a, b = ad[-3], ad[-2]
step = ks[a + 1:b + 1]
t0 = step[0][0]
print(f"step: {len(step)} kernels, {(step[-1][1]-t0)/1e6:.3f} ms")
byq = collections.defaultdict(list)
for k in step: byq[k[3]].append(k)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"  queue {q}: {len(lst):4d} kernels, busy {sum(e-s for s,e,_,_ in lst)/1e6:7.3f} ms, first {(lst[0][0]-t0)/1e6:7.3f} last end {(lst[-1][1]-t0)/1e6:7.3f}")
ev = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
depth = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    depth += d; last = t
print("  resident kernels: " + "  ".join(f"{k}: {v/1e6:.3f} ms" for k, v in sorted(hist.items())))
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]) * 1e6 + t0, float(sys.argv[3]) * 1e6 + t0
    for s, e, n, q in step:
        if e >= lo and s <= hi: print(f"   {(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} us  q{q:>3s}  {n}")
# idle gaps (no kernel resident) of at least 6 us: where, and which kernel ended / started around them
ev = sorted(step, key=lambda k: k[0])
end = step[0][1]; prev = step[0]; prevend = step[0]
gaps = []
for k in ev[1:]:
    if k[0] > end:
        gaps.append((k[0] - end, (end - t0) / 1e3, prevend[2], k[2], k[3]))
    if k[1] > end:
        end = k[1]; prevend = k
big = [g for g in gaps if g[0] >= 6000]
print(f"  idle gaps: {len(gaps)} total {sum(g[0] for g in gaps)/1e6:.3f} ms; >= 6 us: {len(big)} total {sum(g[0] for g in big)/1e6:.3f} ms")
for g in sorted(big, key=lambda g: -g[0])[:14]:
    print(f"     {g[0]/1e3:6.1f} us at {g[1]:8.1f}  after {g[2][:34]:34s} before {g[3][:34]:34s} q{g[4]}")
