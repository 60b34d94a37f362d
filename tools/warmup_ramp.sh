# dev: does the step rate the driver's short run (--steps 20 --warmup 5) sees differ from the steady state?  Same box, alternating.
cd /root/repo
run() { python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("steps", d["steps"], "warmup", d["warmup"], ":", d["ms_per_step"], "ms", d["value"], "clips/s, clock", d["roofline"]["shader_clock_ghz"])'; }
for rep in 1 2; do
  run 20 5; run 20 50; run 20 200; run 100 20; run 300 30
done
