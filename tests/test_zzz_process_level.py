"""Tests that start fresh processes on the GPU (collected last: a process-level fault must not hide the parity tests).

They guard the rule that a HIP graph with parallel branches is instantiated with the hardware queues levelled
(HipOps.level_hw_queues): the HIP runtime's graph launch reads past the end of the graph's internal stream list when
two of those streams share the launch stream's hardware queue, which happens once streams have been destroyed
unevenly - by an earlier engine's graphs, for one - or when the process has fewer than four hardware queues."""
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


def test_forked_graph_after_uneven_stream_destruction():
    """The hazard itself, made deterministic: two streams that share the null stream's hardware queue are destroyed,
    then a graph with four branches is instantiated and replayed from the null stream.  Without the levelling this
    process dies inside hipGraphLaunch; with it the graph runs and writes the right values."""
    p = _run([os.path.join(ROOT, "tools", "diag_hw_queues.py"), "auto"], 120, DIAG_GRAPH="1", DIAG_NO_AFTER="1",
             DIAG_LEVEL="1")
    assert p.returncode == 0 and p.stdout.strip().endswith("done"), (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


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
