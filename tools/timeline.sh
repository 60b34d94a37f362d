# dev: one step under rocprofv3 --kernel-trace: idle gaps (tools/gap_analysis.py) and the per-queue timeline
# (tools/trace_timeline.py [from_ms to_ms] lists the kernels of a window).  usage: timeline.sh [from_ms to_ms]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o kt -- python /root/repo/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra > /tmp/kt2.log 2>&1
f=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1)
cd /root/repo
python tools/gap_analysis.py $f 8 | head -3
python tools/trace_timeline.py $f $1 $2 | head -${3:-120}
