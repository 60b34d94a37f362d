cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_switches.py tests/test_gpu_precision.py -x -q 2>&1 | tail -3
for m in 1 2 0; do echo "== AVID_BS_WIDE=$m"; AVID_BS_WIDE=$m timeout 300 python tools/conv_bench.py 64 2>/dev/null | grep -E 'layer|c3\.|c4\.|c5\.|a\.b[234]|g1152'; done
run() { AVID_BS_WIDE=$1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r.get("mcycles_per_step"), "Mcyc", {k.replace("igemm_pk_kernel","pk"):v["ms_per_step"] for k,v in r["mfma_kernels"].items() if "2,2,2,2" in k})' "BS_WIDE=$1"; }
for rep in 1 2 3; do run 1; run 2; done
