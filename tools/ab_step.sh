# dev: A/B of two library builds on one box: step time and the per-kernel table of bench.py
cd /root/repo
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], "conv kernels", d["roofline"]["all_conv_kernels"]["ms_per_step"]); print("   ", {k.replace("igemm_pk_kernel","pk").replace("_kernel",""):v["ms_per_step"] for k,v in d["roofline"]["mfma_kernels"].items() if v["ms_per_step"] > 0.25}); print("   ", d.get("phases_ms") or "")'
for i in 1 2 3; do
AVID_HIP_LIB=/root/repo/avid-cma_amd/avid_hip/libavid_hip_${1:-nosplit}.so python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" ${1:-nosplit}
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra 2>/dev/null | python -c "$show" default
done
