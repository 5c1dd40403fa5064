"""GPU: the never-yet-executed HIP builders of a batch's relation tensors and relation index (tests/zzz_hip_relbatch_cases.py: 25 cases
against the host builders, array for array, incl. the C2 batch and the device_all loader through the Prefetcher) run in a CHILD process
under a hard timeout.  Code no GPU has run yet must not be able to take the suite -- or the box -- down with it: a fault or a spinning
kernel kills the child only.  Reported as XPASS when all cases pass, XFAIL otherwise (with the child's output); sorted last.  Remove the
indirection and the xfail mark after the first green run (tools/r4_first_call.sh runs the cases directly)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="first execution on an MI355X: written after round 3's GPU minutes were spent")
def test_hip_relation_builders_in_a_child_process():
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "zzz_hip_relbatch_cases.py"), "-m", "gpu", "-q", "--tb=short", "-p",
           "no:cacheprovider"]
    try:
        out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
    except subprocess.TimeoutExpired as e:
        print((e.stdout or b"").decode(errors="replace")[-4000:])
        pytest.fail("the child did not finish in 150 s (killed)")
    text = out.stdout.decode(errors="replace")
    print(text[-6000:])
    try:                                                    # (kept beside the other GPU logs when the box has the directory)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "zzz_hip_relbatch_child.log"), "w") as f:
            f.write(text)
    except OSError:
        pass
    assert out.returncode == 0, "child pytest rc = %d" % out.returncode
