"""The library's results must not depend on what else runs on the GPU (round 4).

Two processes on cuda:0 at once, each repeating the head's training forward and comparing every pass with its first one bit for bit
(tools/cotenancy_stress.py).  Alone on the device the passes always agreed; with a second process, a build whose kernels carried the
compiler's packed fp32 VALU ops deviated in ~1 % of the passes (3000 passes: ~30 events per process).  The product build carries
none (tests/test_cabi_and_host.py::test_device_code_carries_no_packed_fp32_arithmetic); this is the end-to-end watch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stress(mode, n):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cotenancy_stress.py"), mode, str(n)], capture_output=True, text=True,
                       env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("cotenancy_stress")]
    assert len(lines) == 2, r.stdout[-2000:] + r.stderr[-2000:]
    return r.returncode == 0 and all(" 0 of %d deviated" % n in l for l in lines), lines


@pytest.mark.parametrize("mode,n", [("fwd", 3000), ("step", 60), ("task", 25), ("emd", 400)])
def test_results_do_not_depend_on_a_second_process_on_the_gpu(mode, n):
    """STRICT: one deviating pass fails the test.  A build that is affected deviates in ~1 % of the passes (dozens of events in a run
    of the `fwd` mode).  The summary lines -- with the first deviation's description when there is one -- are appended to
    gpurun_out/cotenancy_first_deviation.txt (merged back from the GPU box), so that an event leaves more than a red test behind.
    `emd` covers the one object that carries packed fp32 instructions (hand-written, destinations disjoint from their sources:
    emd.hip); SAMPLENET_AMD_EMD_SCALAR=1 at build time compiles that unit without them (build.py)."""
    ok, lines = _stress(mode, n)
    if not ok:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "cotenancy_first_deviation.txt"), "a") as f:
                f.write("\n".join(lines) + "\n")
        except OSError:
            pass
    assert ok, "\n".join(lines)
