"""profiles/<pmc_summary>.csv (+ the kernel trace of the same run for launch counts) -> profiles/pmc_traffic.json:
HBM bytes per launch per kernel = (2 * FETCH_SIZE + WRITE_SIZE) KiB / launches  (FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is uncalibrated there and taken as is).
Keys are the names the library's HIP-event timers (and bench.py) use: the epilogue variants of igemm_pk_kernel
(its last template argument) are one kernel there, so their bytes and launches are pooled."""
import csv, json, re, sys, collections
summary, trace, out = sys.argv[1], sys.argv[2], sys.argv[3]
calls = collections.Counter()
for r in csv.DictReader(open(trace)):
    calls[r["Kernel_Name"].split("(")[0]] += 1


def short_name(name):
    m = re.match(r"(?:void )?avid::(\w+)(?:<(.*)>)?", name)
    if not m:
        return None
    short = m.group(1)
    if short == "wino2p_kernel":                  # wino2_kernel's default form (V split by the transform): the timers' name
        short = "wino2_kernel"
    if short == "stem_fwd3p_kernel":              # stem_fwd3_kernel's default form (patch split at commit time): likewise
        short = "stem_fwd3_kernel"
    if m.group(2):
        args = [a.strip() for a in m.group(2).split(",")]
        strided = False
        if short == "igemm_pk_kernel" and len(args) >= 7:          # <WM,WN,TM,TN,MODE,STRIDED,EPI>
            strided, args = args[5] == "true", args[:5]
        elif short == "tconv64_kernel" and len(args) >= 2:                # <MODE,EPI>: the timers pool the epilogue variants
            args = args[:1]
        elif short.startswith(("igemm", "stem")) and args[-1] in ("true", "false"):   # trailing bool = STRIDED
            strided, args = args[-1] == "true", args[:-1]
        elif short in ("xmodal_fused_kernel", "xmodal_finish_kernel") and args[-1] in ("true", "false"):   # <CMA>: the timers' names
            short = short.replace("xmodal", "cma") if args[-1] == "true" else short
            args = args[:-1]
        elif short.startswith("wgrad_") and args[-1] in ("true", "false"):            # trailing bool = SPLIT (split-bf16 products): same timer name
            args = args[:-1]
        short += ("<" + ",".join(args) + ">" if args else "") + ("s2" if strided else "")
    return short


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
