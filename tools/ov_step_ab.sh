#!/bin/bash
cd /root/repo
for i in 1 2 3 4; do for ov in 1.0 0.8; do
  echo -n "OV=$ov step: "; AVID_PK_OVERLAP=$ov python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r["mcycles_per_step"], "Mcyc")'
done; done
