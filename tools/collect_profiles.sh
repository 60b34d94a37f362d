set -u
cd /root/repo
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --breakdown > gpurun_out/prof/bench.json 2> gpurun_out/prof/breakdown.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/prof/kt.log 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/prof/kernel_stats.csv
cp $(find /tmp/kt -name "*kernel_trace.csv" | head -1) gpurun_out/prof/kernel_trace.csv
bash tools/pmc.sh gpurun_out/prof/pmc python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof/pmc.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof/pmc/pmc_summary.csv $(ls gpurun_out/prof/pmc/p5/*kernel_trace.csv | head -1) gpurun_out/prof/pmc_traffic.json > /dev/null 2>&1
ls -la gpurun_out/prof
