"""Short runs of round 5's randomised parameter x input x repetition stress scripts (profiles/stress_*.py; DESIGN 4.1 has their long runs:
0 mismatches in 130 k extraction runs, 58 k matcher rounds, 11.8 k prep rounds): every stage against the CPU oracle with PARAMETERS drawn
at random, every configuration repeated (a race shows as a repetition that differs) in both forms of every size-dependent choice."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", script)] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    return out


def test_extraction_parameter_stress_short(gpu):
    out = _run("stress_params.py", 12, 900_000, "exact")
    last = out.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out, out[-3000:]
    assert int(last.split("configurations ")[1].split(",")[0]) >= 50


def test_matcher_parameter_stress_short(gpu):
    out = _run("stress_match_params.py", 10)
    last = out.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out, out[-3000:]
    assert int(last.split("rounds ")[1].split(",")[0]) >= 200


def test_prep_stress_short(gpu):
    out = _run("stress_prep.py", 8)
    last = out.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out, out[-3000:]


# Round 6 (VERDICT r5 weak #3 / item 6 iii): the four scripts that found round 5's LM and facade defects, as short runs (~10 s of
# random configurations each; every script draws its seeds in a fixed order, so a short run repeats the first seeds of the long one)
def test_window_stress_short(gpu):
    out = _run("stress_window.py", 10)
    last = out.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out, out[-3000:]
    assert int(last.split("windows ")[1].split(" ")[0]) >= 5


def test_step_stress_short(gpu):
    out = _run("stress_step.py", 10)
    last = out.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out, out[-3000:]
    assert int(last.split("steps ")[1].split(",")[0]) >= 2


def test_facade_stress_short(gpu):
    out = _run("stress_facade.py", 10)
    last = out.strip().splitlines()[-1]
    # (sizes, iterations, terminations, pair sets of every sweep; sample states 1e-6 - in the default arithmetic a drift below 5e-5 is
    # reported, not failed: DESIGN 4.1)
    assert "drift below 5e-5: 0)" in last, out[-3000:]
    assert int(last.split("streams ")[1].split(",")[0]) >= 1


def test_sharded_window_stress_short(gpu):
    out = _run("stress_sharded_window.py", 10)
    last = out.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out, out[-3000:]
    assert int(last.split("windows ")[1].split(",")[0]) >= 2
