"""The example programs run end to end on the device at toy sizes (each prints one JSON line)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_examples_run():
    d = run("monodomain_fhn.py", "--n", "12", "--steps", "3")
    assert d["ms_per_time_step"] > 0 and d["phi_range"][1] > 0.5
    d = run("monodomain_fhn.py", "--n", "10", "--steps", "3", "--ionic", "tt06")
    assert d["phi_range"][1] > 0.0
    d = run("spiral_wave_2d.py", "--n", "16", "--tend", "20")
    assert d["time_steps"] == 20
    d = run("mechanics_contraction.py", "--n", "4", "--steps", "5")
    assert d["all_converged"] and d["mean_shortening_x"] > 0
    d = run("mechanics_contraction.py", "--n", "4", "--order", "1", "--steps", "10", "--sarcomere", "rdq20", "--tmax", "60")
    assert d["all_converged"]
    d = run("land2015_beam.py")                                  # Newton + Chebyshev-preconditioned CG, all on the device
    assert d["converged"] and abs(d["tip_deflection_z"] - 3.17) <= 0.02
    d = run("electromechanics_lv.py", "--nc", "8", "--nr", "2", "--nl", "4", "--tend", "20", "--mech-every", "5")
    assert d["all_converged"] and d["mechanics_solves"] == 4 and d["final"]["activated_fraction"] > 0.2
