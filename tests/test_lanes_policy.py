"""CPU test of the launch-lane policy (3dscan_amd/csrc/sl3d_lanes.h -- the header the library compiles, free of HIP calls): random call
sequences through the policy and through a model of the context's stream and the two lanes with their events
(tests/native/lanes_policy_check.cpp).  Safety: whatever touched a launch's views before it is ordered before the launch, and every call
that gives the stream work is ordered behind every earlier launch.  Policy: lanes only in a long series, never right behind another call,
never for a launch that repeats the previous launch's views on the stream.  The model has teeth: with one of the plan's waits left out
it reports violations."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "lanes_policy_check.cpp")


def _build(tmp_path, name, *flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", *flags, SRC, "-o", exe])
    return exe


def test_lane_policy_orders_every_dependency(tmp_path):
    exe = _build(tmp_path, "lanes_check")
    for seed in (1, 2, 3):
        p = subprocess.run([exe, "300", str(seed)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr[-2000:]
        words = p.stdout.split()
        launches, on_lanes = int(words[2]), int(words[4].lstrip("("))
        assert launches > 30000 and 0.1 * launches < on_lanes < 0.6 * launches, p.stdout   # (the lanes ARE used, and not for everything)


@pytest.mark.parametrize("mutant", ["DROP_MAIN_WAIT", "DROP_OTHER_WAIT", "DROP_JOIN"])
def test_the_model_objects_when_a_wait_is_left_out(tmp_path, mutant):
    exe = _build(tmp_path, "lanes_" + mutant.lower(), "-D" + mutant)
    p = subprocess.run([exe, "200", "5"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and " 0 violations" not in p.stdout, p.stdout
