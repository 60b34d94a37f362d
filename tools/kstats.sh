#!/bin/bash
# dev tool: rocprofv3 kernel stats of a short bench run -> gpurun_out/kstats.csv
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
(cd /tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kstats.csv
