"""profiles/<pmc_summary>.csv (+ the kernel trace of the same run for launch counts) -> profiles/pmc_traffic.json:
HBM bytes per launch per kernel = (2 * FETCH_SIZE + WRITE_SIZE) KiB / launches  (FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is uncalibrated there and taken as is).
Keys are the names the library's HIP-event timers (and bench.py) use = rocprofv3's names without the namespace: every
kernel under its own template arguments; only the trailing epilogue / pre-split arguments of igemm_pk_kernel and tconv64_kernel
(one kernel per layer shape to the timers) are pooled."""
import csv, json, os, re, sys, collections
summary, trace, out = sys.argv[1], sys.argv[2], sys.argv[3]
calls = collections.Counter()
for r in csv.DictReader(open(trace)):
    calls[r["Kernel_Name"].split("(")[0]] += 1


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import timer_name as short_name      # rocprofv3's name -> the library timers' name


tot_bytes, tot_calls = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(summary)):
    name = r["kernel"].replace(";", ",")
    key = next((k for k in calls if k.replace(" ", "")[:60] == name.replace(" ", "")[:60]), None)
    n = calls.get(key, 0)
    short = short_name(name)
    if not n or short is None:
        continue
    tot_bytes[short] += (2 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024
    tot_calls[short] += n
res = {k: round(tot_bytes[k] / tot_calls[k]) for k in tot_bytes}
# stamp: which kernel sources the counters were taken from (bench.py refuses the table when they differ from the ones
# it runs; there is no git on the GPU box — tools/stamp_traffic.py adds the commit afterwards)
import datetime, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
res["_source"] = {"csrc_sha256": csrc_digest(), "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ")}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
