# dev: A/B of a variant library build (tools/build_variant.sh NAME ...) against the default one, same box, alternating, 300 steps.
# usage: ab_lib.sh NAME [rounds=3] [kernel-name substring to print]
cd /root/repo
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r.get("mcycles_per_step"), "Mcyc", {k.replace("_kernel",""):v["ms_per_step"] for k,v in r["mfma_kernels"].items() if len(sys.argv) > 2 and sys.argv[2] and sys.argv[2] in k})'
for i in $(seq 1 ${2:-3}); do
AVID_HIP_LIB=/root/repo/avid-cma_amd/avid_hip/libavid_hip_$1.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" "$1" "${3:-}"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" default "${3:-}"
done
