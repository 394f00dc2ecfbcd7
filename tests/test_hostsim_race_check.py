"""Data-race check of the kernels without a GPU.

tests/hostsim builds the library's kernels for the host; `make race` builds them once more under
clang's thread sanitizer, every simulated GPU thread being a fiber of its own to the detector and
only barriers, cross-lane operations and workgroup start / end ordering them.  race_selftest holds
the controls (a missing barrier and a plain read-modify-write of one device word by many
workgroups must be reported; the barrier, a wave-level operation and an atomic must not), and
race_driver calls every entry point of the C ABI on small inputs.  Any report fails the test."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim")
BIN = os.path.join(SIM, "_build", "tsan")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", SIM, "-j8", "race"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def run(exe, args, extra_env=None):
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=0 exitcode=66 external_symbolizer_path=/opt/rocm/lib/llvm/bin/llvm-symbolizer"
    env.update(extra_env or {})
    r = subprocess.run([os.path.join(BIN, exe)] + args, capture_output=True, text=True, env=env, timeout=900)
    return r.returncode, r.stdout + r.stderr


@pytest.mark.parametrize("mode,racy", [("lds-race", True), ("lds-barrier", False), ("lds-wave", False),
                                       ("global-race", True), ("global-atomic", False)])
def test_detector_controls(built, mode, racy):
    # The device-memory control only races when several worker threads of the simulation take
    # workgroups (those that follow each other on one worker count as ordered): on a loaded
    # machine one worker can finish all of them before the others start, so a racy control that
    # stays silent is run again before it counts as a failure of the detector.
    for attempt in range(3 if racy else 1):
        code, out = run("race_selftest", [mode])
        reported = "WARNING: ThreadSanitizer: data race" in out
        if reported == racy:
            break
    assert reported == racy, out[-2000:]
    assert code == (66 if racy else 0)


@pytest.mark.parametrize("args,env", [
    (["320", "200", "400", "synth"], {}),
    (["320", "200", "400", "noise"], {"ORBX_OCT_KEYS_LDS": "0", "ORBX_GUIDED_GATHER": "1"}),
    (["200", "160", "300", "faint"], {}),
    (["320", "200", "400", "synth"], {"ORBX_FAST_SINGLE_PASS": "1", "HOSTSIM_ORDER": "reverse",
                                      "ORBX_GUIDED_GATHER": "0"}),
])
def test_no_race_in_any_entry_point(built, args, env):
    code, out = run("race_driver", args, env)
    assert "ThreadSanitizer" not in out, out[-6000:]
    assert code == 0, out[-3000:]
    assert "race_driver: done" in out


def test_host_code_under_address_and_undefined_behaviour_sanitizers():
    """The same driver with AddressSanitizer, LeakSanitizer and UBSan on every translation unit (the
    library's host code included, and the kernels as compiled for the host): no report, nothing
    leaked once the handles are destroyed."""
    r = subprocess.run(["make", "-C", SIM, "-j8", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(SIM, "_build", "asan", "race_driver"), "320", "200", "400", "synth"],
                         capture_output=True, text=True, env=env, timeout=900)
    text = out.stdout + out.stderr
    assert out.returncode == 0 and "ERROR" not in text and "runtime error" not in text, text[-4000:]
    assert "race_driver: done" in text
