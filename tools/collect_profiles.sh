# Round profile collection on the GPU box (run through gpurun): bench line + HIP-event breakdown, rocprofv3 kernel
# stats of the same command, the PMC passes (tools/pmc.sh: separate --pmc runs as MI355X_MICROARCH.md prescribes)
# and the per-kernel HBM traffic table.  Only the summaries are kept (the traces are hundreds of MB):
#   gpurun_out/prof/{bench.json, breakdown.txt, kernel_stats.csv, pmc_summary.csv, pmc_traffic.json}
# Copy them to profiles/<round>_* and pmc_traffic.json to profiles/pmc_traffic.json (bench.py reads that one and
# reports roofline.traffic = null, with a note on stderr, if its dominant kernel is missing from it).
set -u
cd /root/repo
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
# Order: the PMC passes and the traffic table FIRST — stamped with the digest of the kernel sources and put where
# bench.py reads it — the kernel trace next, bench.py LAST, so that the bench line kept under profiles/ carries roofline.traffic.
# (per-kernel durations only mean something while kernels do not share the chip: the trace and the counter passes run with
# the weight gradients on the compute streams and the two towers on one stream (AVID_DEFER_WGRAD=0 AVID_OVERLAP_TOWERS=0), like
# bench.py's own HIP-event pass; `value` in bench.json is measured with the trailing streams on)
rm -rf /tmp/pmc_out
AVID_DEFER_WGRAD=0 AVID_OVERLAP_TOWERS=0 bash tools/pmc.sh /tmp/pmc_out python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof/pmc.log 2>&1
cp /tmp/pmc_out/pmc_summary.csv gpurun_out/prof/pmc_summary.csv
python tools/pmc_traffic.py /tmp/pmc_out/pmc_summary.csv $(ls /tmp/pmc_out/p5/*/*kernel_trace.csv /tmp/pmc_out/p5/*kernel_trace.csv 2>/dev/null | head -1) gpurun_out/prof/pmc_traffic.json > /dev/null 2>> gpurun_out/prof/pmc.log
cp gpurun_out/prof/pmc_traffic.json profiles/pmc_traffic.json      # (this box's copy of the repo: what the bench run below reads)
(cd /tmp && rm -rf /tmp/kt && AVID_DEFER_WGRAD=0 AVID_OVERLAP_TOWERS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /root/repo/gpurun_out/prof/kt.log 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/prof/kernel_stats.csv
python bench.py --steps 20 --warmup 5 --breakdown > gpurun_out/prof/bench.json 2> gpurun_out/prof/breakdown.txt
tail -c 2000 gpurun_out/prof/pmc.log > gpurun_out/prof/pmc_tail.log; rm -f gpurun_out/prof/pmc.log gpurun_out/prof/kt.log
ls -la gpurun_out/prof
