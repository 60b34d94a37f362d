cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_precision.py -x -q -k "stem" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_switches.py -x -q -k "STEM" 2>&1 | tail -3
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r.get("mcycles_per_step"), "Mcyc", {k:v["ms_per_step"] for k,v in r["mfma_kernels"].items() if "stem_fwd3" in k})'
for i in 1 2 3; do
for e in "AVID_STEM_FWD_PRE=0" "AVID_STEM_FWD_TM=1" "AVID_STEM_FWD_TM=2"; do
env $e python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" "$e"
done; done
