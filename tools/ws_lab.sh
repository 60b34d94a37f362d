#!/bin/bash
# Development (GPU box): the weight-stationary order of igemm_pk_kernel's K-split tails (AVID_PK_WS 0 / 1) — per layer, per step,
# and in counter traffic.
cd /root/repo
for ws in 0 1 0 1; do
  for layer in c5.spt_s2 c5.spt c5.tmp_s2 a.b3_s2 a.b3 a.b4a a.b4; do
    echo -n "PK_WS=$ws "; AVID_PK_WS=$ws python tools/conv_bench.py 64 $layer 2>/dev/null | tail -1 | cut -c1-100
  done
done
for ws in 0 1 0 1; do
  echo -n "PK_WS=$ws step: "; AVID_PK_WS=$ws python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r["mcycles_per_step"], "Mcyc", {k.replace("igemm_pk_kernel","pk"): v["ms_per_step"] for k, v in r["mfma_kernels"].items() if "igemm_pk" in k})'
done
bash tools/traffic_ab.sh "igemm_pk|wino_kernel|splitk" AVID_PK_WS=0
bash tools/traffic_ab.sh "igemm_pk|wino_kernel|splitk" AVID_PK_WS=1
