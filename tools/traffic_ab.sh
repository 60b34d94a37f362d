#!/bin/bash
# usage: tools/traffic_ab.sh <kernel regex> [ENV=v ...] — HBM-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's
# gfx950 correction) of every kernel whose name matches, in a short single-stream bench run with the given environment:
# two rocprofv3 --pmc passes.  For A/Bs of one switch on one box (the full table: tools/collect_profiles.sh).
K=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/ta1 /tmp/ta2
env AVID_DEFER_WGRAD=0 AVID_OVERLAP_TOWERS=0 "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ta1 -o p -- python $R/bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > /tmp/ta1.log 2>&1
env AVID_DEFER_WGRAD=0 AVID_OVERLAP_TOWERS=0 "$@" rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ta2 -o p -- python $R/bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > /tmp/ta2.log 2>&1
python - "$K" "$*" <<'PY'
import csv, glob, sys, re, collections
sys.path.insert(0, "/root/repo/tools")
from kernel_names import timer_name
pat = re.compile(sys.argv[1])
tot = {"FETCH_SIZE": collections.Counter(), "WRITE_SIZE": collections.Counter()}
cnt = {"FETCH_SIZE": collections.Counter(), "WRITE_SIZE": collections.Counter()}
for d, name in (("/tmp/ta1", "FETCH_SIZE"), ("/tmp/ta2", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = timer_name(r["Kernel_Name"])
            if k and pat.search(k) and r["Counter_Name"] == name:
                tot[name][k] += float(r["Counter_Value"]); cnt[name][k] += 1
print("traffic [%s]" % sys.argv[2])
for k in sorted(tot["FETCH_SIZE"]):
    f = tot["FETCH_SIZE"][k] / max(cnt["FETCH_SIZE"][k], 1) * 1024 / 1e6
    w = tot["WRITE_SIZE"][k] / max(cnt["WRITE_SIZE"][k], 1) * 1024 / 1e6
    print(f"  {k:34s} launches {cnt['FETCH_SIZE'][k]:4d}  2xFETCH {2 * f:8.1f} MB  WRITE {w:7.1f} MB  total {2 * f + w:8.1f} MB / launch")
PY
