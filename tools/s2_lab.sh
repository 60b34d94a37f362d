#!/bin/bash
# Development (GPU box): strided input gradients of the Cin % 128 == 0 layers on the 128 x 128 tile (AVID_S2_WIDE=1) or the
# 128 x 64 tile with pre-split weights (0) — per layer and per step.
cd /root/repo
for w in 1 0 1 0; do
  for layer in c3.tmp_s2 c4.spt_s2 c4.tmp_s2 c5.spt_s2 c5.tmp_s2 a.b3_s2; do
    echo -n "S2_WIDE=$w "; AVID_S2_WIDE=$w python tools/conv_bench.py 64 $layer 2>/dev/null | tail -1 | cut -c1-150
  done
done
for w in 1 0 1 0; do
  echo -n "S2_WIDE=$w step: "; AVID_S2_WIDE=$w python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r["mcycles_per_step"], "Mcyc", {k.replace("igemm_pk_kernel","pk"): v["ms_per_step"] for k, v in r["mfma_kernels"].items() if "s2" in k or "wino" in k})'
done
