#!/bin/bash
# usage: tools/pmc.sh <outdir> <cmd...>   — runs the separate rocprofv3 --pmc passes (MI355X_MICROARCH.md slots)
set -u
OUT=$1; shift
export TMPDIR=/tmp
R=$PWD
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac      # absolute or relative to the caller's directory
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- "$@" > $OUT/p$i.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{out}/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "WRITE_SIZE"):
            pass
names = sorted({c for v in agg.values() for c in v})
with open(f"{out}/pmc_summary.csv", "w") as fo:
    fo.write("kernel," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        fo.write(k.replace(",", ";") + "," + ",".join(f"{v.get(n, 0):.0f}" for n in names) + "\n")
print(open(f"{out}/pmc_summary.csv").read()[:6000])
PY
