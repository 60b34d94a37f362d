#!/bin/bash
# Development (GPU box): wino_wgrad_kernel's time with parts of it switched off (AVID_WW_DBG variants built by
# tools/build_variant.sh) and the per-wave cycle trace, on the conv2x / conv3x / conv4x spatial layers.
cd /root/repo
L=/root/repo/avid-cma_amd/avid_hip
for v in "" _ww1 _ww2 _ww3 _ww4 _ww7 _ww8 _ww16; do
  lib=$L/libavid_hip$v.so
  [ -f $lib ] || continue
  for layer in c2.spt c3.spt c4.spt; do
    echo -n "variant${v:-_default} $layer: "
    AVID_HIP_LIB=$lib CB_VERBOSE=1 python tools/conv_bench.py 64 $layer 2>/dev/null | grep "bwd wino_wgrad_kernel" | awk '{print $3, $4}'
  done
done
[ -f $L/libavid_hip_wwtrace.so ] && AVID_HIP_LIB=$L/libavid_hip_wwtrace.so python tools/wino_wgrad_trace.py 2>/dev/null
