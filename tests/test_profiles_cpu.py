"""The committed profile summaries the bench line and the docs cite must describe the tree that is built.  profiles/MANIFEST.json lists,
per summary, the commit it was measured on and the SOURCES whose kernels it describes; a summary is stale -- and this suite red -- as
soon as one of those sources DIFFERS (by content) from what it was at that commit (round 5's guard tolerated 12 later csrc commits and did not look at the detector
summaries at all).  Every kernel the EKF summaries name must also be a symbol of the built librekf.so."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFEST = os.path.join(ROOT, "profiles", "MANIFEST.json")


def _symbols(lib):
    return subprocess.run(["nm", "-D", "-C", os.path.join(ROOT, "reflector_ekf_slam_amd", lib)], capture_output=True, text=True, check=True).stdout


def _norm(name):
    return re.sub(r"\s+", "", name.replace("void ", ""))


def test_manifest_covers_what_the_bench_line_reads():
    man = json.load(open(MANIFEST))["summaries"]
    for f in ("kernel_avg_us.json", "pmc_downdate.json", "pmc_mfma.json"):
        assert f in man, f"{f} is read by bench.py but profiles/MANIFEST.json does not list it"
    assert any("det3d.hip" in " ".join(e["sources"]) for e in man.values()), "no detector summary in the manifest"
    for f, e in man.items():
        assert os.path.exists(os.path.join(ROOT, "profiles", f)), f"{f} is in the manifest but not in profiles/"
        assert e.get("commit") and e.get("sources"), f
        for src in e["sources"]:
            assert os.path.exists(os.path.join(ROOT, src)), (f, src)
        j = os.path.join(ROOT, "profiles", f)
        if f.endswith(".json"):
            c = json.load(open(j)).get("_commit")
            assert c is None or c == e["commit"], f"{f} says it was measured on {c}, the manifest on {e['commit']}"


def test_every_profiled_kernel_is_in_the_built_library():
    syms = re.sub(r"\s+", "", _symbols("librekf.so"))
    avg = json.load(open(os.path.join(ROOT, "profiles", "kernel_avg_us.json")))
    ours = [k for k in avg if not k.startswith("_") and not k.startswith("__amd")]
    assert any("k_mid<4,0>" in _norm(k) for k in ours)
    for k in ours:
        assert _norm(k) in syms, f"profiles/kernel_avg_us.json names {k!r}, which the built librekf.so does not export"
    for f in ("pmc_downdate.json", "pmc_mfma.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", f)))
        assert _norm(j["kernel"].split("(")[0]) in syms, (f, j["kernel"])


def test_no_summary_is_older_than_the_sources_it_describes():
    """Stale = a described source DIFFERS from what the summary was measured on: the blob of every source at the manifest's commit against
    the file in the working tree (content, not history: a change that was committed and reverted leaves the measured code in place)."""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (a GPU box snapshot)")
    man = json.load(open(MANIFEST))["summaries"]
    stale = []
    for f, e in man.items():
        c = e["commit"]
        assert subprocess.run(["git", "merge-base", "--is-ancestor", c, "HEAD"], cwd=ROOT).returncode == 0, (f, c)
        for src in e["sources"]:
            then = subprocess.run(["git", "rev-parse", f"{c}:{src}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()
            now = subprocess.run(["git", "hash-object", src], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()
            if then != now:
                stale.append((f, c, src))
    assert not stale, f"stale profile summaries (re-run scripts/gpu_profile_round.sh and scripts/profiles_commit.py): {stale}"
