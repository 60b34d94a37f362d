# dev: the whole step under the driver's frequency policies (the clock the chip holds follows the densest kernels).
# auto (the default the bench is quoted on) / high / perf-determinism at two caps, alternating; resets to auto at the end.
cd /root/repo
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms", d["value"], "clips/s, in-bench clock", d["roofline"]["shader_clock_ghz"], "GHz")' "$1"; }
rocm-smi --showperflevel --showclocks 2>&1 | grep -v '^=' | head -20
for rep in 1 2; do
  rocm-smi --setperflevel auto > /dev/null 2>&1; run "auto"
  rocm-smi --setperflevel high 2>&1 | grep -iE 'error|success|level' | head -2; run "high"
done
for mhz in 2100 2400; do
  rocm-smi --setperfdeterminism $mhz 2>&1 | grep -iE 'error|success|determinism' | head -2; run "determinism_$mhz"
done
rocm-smi --resetperfdeterminism > /dev/null 2>&1; rocm-smi --setperflevel auto > /dev/null 2>&1
run "auto_again"
