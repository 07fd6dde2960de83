"""The committed profile summaries bench.py reads (profiles/kernel_avg_us.json, pmc_downdate.json, pmc_mfma.json) must describe the
tree that is built: every kernel they name is a symbol of librekf.so, and the commit they were measured on is an ancestor of HEAD with at
most a few later commits touching csrc/ (round 4's summaries named kernels the shipped library no longer had)."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_LATER_CSRC_COMMITS = 12


def _symbols():
    out = subprocess.run(["nm", "-D", "-C", os.path.join(ROOT, "reflector_ekf_slam_amd", "librekf.so")], capture_output=True, text=True, check=True).stdout
    return out


def _norm(name):
    return re.sub(r"\s+", "", name.replace("void ", ""))


def test_every_profiled_kernel_is_in_the_built_library():
    syms = re.sub(r"\s+", "", _symbols())
    avg = json.load(open(os.path.join(ROOT, "profiles", "kernel_avg_us.json")))
    ours = [k for k in avg if not k.startswith("_") and not k.startswith("__amd")]
    assert any("k_mid<4,0>" in _norm(k) for k in ours)
    for k in ours:
        assert _norm(k) in syms, f"profiles/kernel_avg_us.json names {k!r}, which the built librekf.so does not export"
    for f in ("pmc_downdate.json", "pmc_mfma.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", f)))
        assert _norm(j["kernel"].split("(")[0]) in syms, (f, j["kernel"])


def test_the_profiles_were_taken_close_to_head():
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (a GPU box snapshot)")
    for f in ("kernel_avg_us.json", "pmc_downdate.json", "pmc_mfma.json"):
        c = json.load(open(os.path.join(ROOT, "profiles", f))).get("_commit")
        assert c, f"{f} does not say which commit it was measured on"
        assert subprocess.run(["git", "merge-base", "--is-ancestor", c, "HEAD"], cwd=ROOT).returncode == 0, (f, c)
        later = subprocess.run(["git", "rev-list", "--count", f"{c}..HEAD", "--", "reflector_ekf_slam_amd/csrc"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
        assert int(later) <= MAX_LATER_CSRC_COMMITS, f"{f} is {later.strip()} csrc commits behind HEAD: re-run scripts/gpu_profile_round.sh"
