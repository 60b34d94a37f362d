# dev: A/B of an environment setting against the default on one box.  usage: ab_env.sh "env NAME=VALUE" [rounds]
cd /root/repo
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in $(seq 1 ${2:-3}); do
$1 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" "$1"
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" default
done
