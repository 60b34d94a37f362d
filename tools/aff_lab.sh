#!/bin/bash
# Development (GPU box): conv2x's temporal layers applying their input's BatchNorm (AVID_IN_AFFINE 1 / 0) — the step, alternating.
cd /root/repo
for v in 1 0 1 0; do
  echo -n "IN_AFFINE=$v step: "; AVID_IN_AFFINE=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --breakdown 2>/tmp/bd.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], "ms", r["shader_clock_ghz"], "GHz", r["mcycles_per_step"], "Mcyc", {k: v["ms_per_step"] for k, v in r["mfma_kernels"].items() if "tconv" in k or "twgrad" in k})'
  grep -E "^bn_apply_kernel|^timed kernels" /tmp/bd.txt
done
