"""The committed measurement evidence has to belong to the tree it is committed with.  No GPU here — these
checks run in the build container, before a commit, so that a kernel change made after the last `tools/collect_profiles.sh`
fails HERE and not as a stderr line on the driver's box (round 5 ended that way: `roofline.traffic` null in the driver's line).

Rule the checks encode: the LAST GPU action of a round is tools/collect_profiles.sh at HEAD, its summaries are copied to
profiles/<round>_* (and pmc_traffic.json to profiles/pmc_traffic.json), and no file under avid-cma_amd/csrc/ or include/ changes
afterwards.  (This file sorts last on purpose: with `pytest -x` a stale table must not hide what the other tests say.)"""
import csv
import glob
import json
import os
import re
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def _latest_set():
    """(tag, bench.json path, kernel_stats.csv path) of the newest profile set, e.g. 'r06_b'."""
    sets = sorted(re.match(r"(r\d+_\w+?)_bench\.json$", os.path.basename(p)).group(1)
                  for p in glob.glob(os.path.join(REPO, "profiles", "r*_bench.json"))
                  if re.match(r"r\d+_[a-z]_bench\.json$", os.path.basename(p)))
    tag = sets[-1]
    return tag, os.path.join(REPO, "profiles", tag + "_bench.json"), os.path.join(REPO, "profiles", tag + "_kernel_stats.csv")


def test_traffic_table_was_collected_from_these_kernel_sources():
    """profiles/pmc_traffic.json (what bench.py's roofline.traffic is read from) is stamped with the sha256 of the kernel sources
    its counters were collected from; bench.py reports null when the stamp is not the digest of the sources it runs."""
    import bench
    table = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    stamp = (table.get("_source") or {}).get("csrc_sha256")
    assert stamp == bench.csrc_digest(), (
        "profiles/pmc_traffic.json was collected from other kernel sources than the ones in this tree "
        f"({str(stamp)[:12]} vs {bench.csrc_digest()[:12]}): run tools/collect_profiles.sh on the GPU box (gpurun) and commit "
        "gpurun_out/prof/* as profiles/<round>_* + profiles/pmc_traffic.json — after the LAST change under csrc/ / include/")


def test_latest_profile_set_names_every_kernel_the_bench_line_names():
    """The driver line's mfma_kernels table, profiles/pmc_traffic.json and the rocprofv3 summary of the same round agree key for
    key: every kernel the library's timers name is a kernel rocprofv3 saw (tools/kernel_names.py maps the one to the other)."""
    from kernel_names import timer_name
    tag, bench_json, stats_csv = _latest_set()
    if tag < "r06":
        pytest.skip("profile sets before round 6 pooled several kernels under one timer name")
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    seen = {timer_name(r["Name"]) for r in csv.DictReader(open(stats_csv))} - {None}
    named = set(line["roofline"]["mfma_kernels"])
    assert named <= seen, (tag, sorted(named - seen))
    assert line["roofline"]["kernel"] in seen
    # ... and the record carries the counter traffic of its dominant kernel, from a table that was not stale when it ran
    assert line["roofline"]["traffic"] is not None and not line["roofline"]["traffic_source"].get("stale"), tag
    table = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    assert named <= set(table), sorted(named - set(table))


def test_forward_record_has_no_fraction_above_one():
    """r2p1d_forward: `frac` is the fraction of the issued instructions' peak; the direct-form figure is a
    throughput, not a roofline fraction (1.037 under the name `frac`)."""
    tag, bench_json, _ = _latest_set()
    if tag < "r06":
        pytest.skip("before the rename")
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    fwd = (line.get("extra") or {}).get("r2p1d_forward") or line["roofline"].get("r2p1d_forward")
    assert fwd is not None and 0.0 < fwd["frac"] <= 1.0 and fwd["frac"] == fwd["frac_of_issued_peak"]
    assert "frac" not in fwd.get("direct_form_equivalent", {})
