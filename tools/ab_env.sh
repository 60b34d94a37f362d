# dev: A/B of an environment setting against the default on one box, alternating runs.
# usage: ab_env.sh "env NAME=VALUE" [rounds=3] [steps=300] [kernel-name substring to print]
cd /root/repo
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], "ms", d["value"], "clips/s", r["shader_clock_ghz"], "GHz", r.get("mcycles_per_step"), "Mcyc", {k.replace("_kernel",""):v["ms_per_step"] for k,v in r["mfma_kernels"].items() if len(sys.argv) > 2 and sys.argv[2] and sys.argv[2] in k})'
for i in $(seq 1 ${2:-3}); do
$1 python bench.py --steps ${3:-300} --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" "$1" "${4:-}"
python bench.py --steps ${3:-300} --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" default "${4:-}"
done
