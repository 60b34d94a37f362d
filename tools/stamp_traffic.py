"""Build-box step after tools/collect_profiles.sh: add the commit whose kernel sources profiles/pmc_traffic.json was
collected from (refuses if the table's source digest is not the working tree's)."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import csrc_digest
f = os.path.join(REPO, "profiles", "pmc_traffic.json")
t = json.load(open(f))
src = t.get("_source") or {}
if src.get("csrc_sha256") != csrc_digest():
    sys.exit("profiles/pmc_traffic.json was not collected from this tree's kernel sources")
src["commit"] = subprocess.check_output(["git", "-C", REPO, "log", "-1", "--format=%h", "--", "avid-cma_amd/csrc", "include"], text=True).strip()
t["_source"] = src
json.dump(t, open(f, "w"), indent=1, sort_keys=True)
print(src)
