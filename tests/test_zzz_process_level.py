"""Tests that start fresh processes on the GPU (collected last: a process-level fault must not hide the parity tests).

They guard the rule that this package never destroys a HIP stream, hence never a HIP graph with parallel branches
(engine._PARKED_FORKED_GRAPHS): destroying a stream lets the runtime's signal thread write into the freed stream
object, and uneven destruction makes the runtime's graph launch read past the end of a graph's internal stream list -
as does a process with fewer than four hardware queues, where such graphs are not made at all."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout, **env):
    e = dict(os.environ, PYTHONFAULTHANDLER="1", **env)
    p = subprocess.run([sys.executable, "-u"] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)
    return p


def _stress_ok(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("identity")]
    assert p.returncode == 0 and len(lines) == 2 and all(ln.split()[2] == "same" for ln in lines), \
        (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


def test_graph_runs_in_fresh_processes_equal_eager_runs():
    """The drop-in loop as the first training runs of a process, without and then with the identity extension: HIP
    graphs on several streams against eager launches on one stream, metrics and final weights bit for bit
    (tools/stress_first_run.py: four trainings per process, so the later ones capture their graphs after an earlier
    engine has released its own)."""
    for _ in range(3):
        _stress_ok(_run([os.path.join(ROOT, "tools", "stress_first_run.py")], 280, STRESS_EPOCHS="5"))


def test_fewer_hardware_queues_than_graph_branches():
    """GPU_MAX_HW_QUEUES=2: a forked graph cannot be launched safely, so the step is captured on one stream (with a
    warning) - no crash, same bits as the eager run."""
    p = _run([os.path.join(ROOT, "tools", "stress_first_run.py")], 280, STRESS_EPOCHS="3", GPU_MAX_HW_QUEUES="2")
    _stress_ok(p)
    assert "captured on one stream" in p.stderr


def test_more_hardware_queues_than_four():
    """GPU_MAX_HW_QUEUES=8: forked graphs would run (slower than one stream: their branches are time-sliced), so the
    step is captured on one stream (with a warning) - same bits as the eager run."""
    p = _run([os.path.join(ROOT, "tools", "stress_first_run.py")], 280, STRESS_EPOCHS="3", GPU_MAX_HW_QUEUES="8")
    _stress_ok(p)
    assert "captured on one stream" in p.stderr
