"""Development: VGPR liveness over a kernel's final ISA (CFG-aware), to find where the register allocator's pressure
peaks — `hipcc -S --cuda-device-only`, cut one kernel out (awk '/^<mangled>:/,/s_endpgm/'), then
    python tools/asm_liveness.py kernel.s [bucket]
prints the maximum number of live VGPRs per `bucket` lines and the line of the overall peak."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
bucket = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pref = sys.argv[3] if len(sys.argv) > 3 else 'v'
def regs(tok):
    out = []
    for m in re.finditer(r'(?<![a-z_0-9])%s\[(\d+):(\d+)\]|(?<![a-z_0-9\[:])%s(\d+)\b' % (pref, pref), tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        elif m.group(3): out.append(int(m.group(3)))
    return out
# instructions and basic blocks
ins = []          # (line, op, defs, uses, target or None, falls_through)
label_at = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: label_at[m.group(1)] = len(ins)
    c = l.split(';')[0].strip()
    if not c or c.startswith('.') or c.endswith(':'): continue
    parts = c.split(None, 1)
    op = parts[0]; args = parts[1] if len(parts) > 1 else ''
    ops = [x.strip() for x in args.split(',')] if args else []
    tgt = None; fall = True
    if op.startswith('s_cbranch') or op == 's_branch':
        tgt = ops[0]; fall = op != 's_branch'; d = []; s = []
    elif op == 's_endpgm':
        fall = False; d = []; s = []
    elif op.startswith(('buffer_store', 'ds_write', 'scratch_store', 'global_store', 's_', 'v_cmp_', 'v_cmpx', 'buffer_wbl2', 'buffer_inv')):
        d = []; s = ops
    else:
        d = ops[:1]; s = ops[1:]
    D = set(); S = set()
    for x in d: D.update(regs(x))
    for x in s: S.update(regs(x))
    if op.startswith('v_mfma') or op.startswith('v_accvgpr_write'): pass
    ins.append((i + 1, op, D, S, tgt, fall))
n = len(ins)
succ = [[] for _ in range(n)]
for k, (ln, op, D, S, tgt, fall) in enumerate(ins):
    if fall and k + 1 < n: succ[k].append(k + 1)
    if tgt is not None and tgt in label_at and label_at[tgt] < n: succ[k].append(label_at[tgt])
live_in = [set() for _ in range(n)]
changed = True
it = 0
while changed and it < 50:
    changed = False; it += 1
    for k in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[k]: out |= live_in[s_]
        new = (out - ins[k][2]) | ins[k][3]
        if new != live_in[k]:
            live_in[k] = new; changed = True
b = collections.defaultdict(int)
peak = (0, 0)
for k in range(n):
    v = len(live_in[k]); ln = ins[k][0]
    b[ln // bucket * bucket] = max(b[ln // bucket * bucket], v)
    if v > peak[0]: peak = (v, ln)
print(' '.join('%d:%d' % (k, b[k]) for k in sorted(b)))
print('peak', peak, lines[peak[1] - 1].strip()[:100])
if len(sys.argv) > 4:
    k = [j for j in range(n) if ins[j][0] == peak[1]][0]
    L = sorted(live_in[k])
    print('live at peak:', L)
    # for each live reg: the nearest previous def line and op
    for r in L:
        for j in range(k - 1, -1, -1):
            if r in ins[j][2]:
                print(r, ins[j][0], lines[ins[j][0] - 1].strip()[:80]); break
