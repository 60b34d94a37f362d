# dev: per-layer timings (tools/conv_bench.py) of library variants on one box.  usage: ab_layers.sh "variant ..." "layer ..."
cd /root/repo
for v in default $1; do
  echo "== variant $v"
  if [ "$v" != default ]; then export AVID_HIP_LIB=/root/repo/avid-cma_amd/avid_hip/libavid_hip_$v.so; else unset AVID_HIP_LIB; fi
  for l in $2; do python tools/conv_bench.py 64 $l 2>&1 | tail -1; done
done
