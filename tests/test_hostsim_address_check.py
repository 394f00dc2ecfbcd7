"""Address and convergence check of the kernels without a GPU.

tests/hostsim compiles the library's .hip sources for the host and runs every kernel thread
by thread: device allocations and dynamic LDS sit against guard pages, so an out-of-bounds
access of any kernel stops the run; lanes of a wave that meet at different cross-lane
operations are reported; and because the simulated kernels execute the same source, their
outputs can be compared with the CPU oracle.  Here the whole GPU-marked suite is run against
that build three times: with the guard page behind every allocation; with the guard page in front,
the threads of every workgroup scheduled from the last to the first and every undefined-behaviour
check compiled in as a trap; and with waves and lanes in a fresh pseudo-random order at every
sweep of the scheduler (a result that depends on which thread reaches a point first - a missing
barrier, an order-dependent use of atomics - would change).

This is a check of the kernels' addressing and logic.  It is not a GPU parity result (the
GPU tests proper are the ones marked `gpu`, run on the MI355X) and nothing in the package
loads the simulation build.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("guard,order,lib", [("end", "forward", ""), ("start", "reverse", "ubsan"),
                                             ("end", "shuffle:7", "")])
def test_gpu_suite_under_host_simulation(guard, order, lib):
    # lib "ubsan": the simulation built with every undefined-behaviour check (signed overflow,
    # shifts, misaligned or null accesses ...) compiled in as a trap
    env = dict(os.environ, ORBX_HOSTSIM="1", HOSTSIM_GUARD=guard, HOSTSIM_ORDER=order, ORBX_HOSTSIM_LIB=lib)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-q", "-x",
           "-p", "no:faulthandler", "-p", "no:cacheprovider"]
    if (guard, order) == ("end", "forward"):
        # every schedule knob once, with the parts joined at the end of the call (the pipelined half
        # of that matrix runs on the GPU and, one case of each kind, in the other two variants)
        cmd += ["-k", "not schedule_knobs or False-ORBX"]
    else:
        cmd += ["-k", "not schedule_knobs or True-ORBX_PARTS-3 or True-ORBX_FAST_QCAP"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and " failed" not in out.stdout, tail


def test_random_configurations_under_host_simulation():
    """tools/fuzz_parity.py draws image sizes, contents, extractor parameters and matcher inputs
    and compares every stage with the oracle; a short fixed range of its cases runs here (longer
    ranges are run by hand: 1600 cases passed when this was written)."""
    env = dict(os.environ, ORBX_HOSTSIM="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--cases", "40", "--start",
                          "5000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "40 cases, 0 failures" in out.stdout, tail
