"""Dev tool: from a rocprofv3 --kernel-trace csv, how busy is the GPU inside the timed steps?  Reports, over the
last N launches of adam_flat_kernel (one per step): step period, union-busy time, idle time, the idle time split by
the kernel that FOLLOWS each gap, and the longest gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
adam = [e for e in ev if "adam_flat" in e[2]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t0, t1 = adam[-nsteps - 1][1], adam[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy = 0; cur_s, cur_e = None, None
gaps = []
for s, e, n in win:
    if cur_e is None: cur_s, cur_e = s, e; continue
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"{nsteps} steps: period {span/nsteps/1e6:.3f} ms; busy {busy/nsteps/1e6:.3f} ms; idle {(span-busy)/nsteps/1e6:.3f} ms "
      f"({100*(span-busy)/span:.1f} %); kernels/step {len(win)/nsteps:.0f}; sum of kernel durations {sum(e-s for s,e,_ in win)/nsteps/1e6:.3f} ms")
by = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    k = n.split("(")[0][-60:]
    by[k][0] += g; by[k][1] += 1
print("idle time by the kernel that ends the gap (us/step, count/step, avg us):")
for k, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {k:62s} {g/nsteps/1e3:8.1f} {c/nsteps:6.1f} {g/c/1e3:6.2f}")
dur = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    k = n.split("(")[0][-60:]
    dur[k][0] += e - s; dur[k][1] += 1
print("small kernels (avg < 12 us): us/step, count/step, avg us")
for k, (d, c) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    if d / c < 12e3: print(f"  {k:62s} {d/nsteps/1e3:8.1f} {c/nsteps:6.1f} {d/c/1e3:6.2f}")
