#!/bin/bash
# Development (GPU box): the planner's price of two co-resident tail units (AVID_PK_OVERLAP) — per layer and per step.
cd /root/repo
for ov in 1.0 0.8 0.6 1.0 0.8 0.6; do
  for layer in c4.tmp c4.res c5.spt_s2 c5.spt c5.tmp c5.tmp_s2 a.b2_s2 a.b2 a.b3_s2 a.b3 a.b4a a.b4; do
    echo -n "OV=$ov "; AVID_PK_OVERLAP=$ov python tools/conv_bench.py 64 $layer 2>/dev/null | tail -1 | cut -c1-100
  done
done
for ov in 1.0 0.8 0.6 1.0 0.8 0.6; do
  echo -n "OV=$ov step: "; AVID_PK_OVERLAP=$ov python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r["mcycles_per_step"], "Mcyc", {k.replace("igemm_pk_kernel","pk"): v["ms_per_step"] for k, v in r["mfma_kernels"].items() if "igemm_pk" in k})'
done
