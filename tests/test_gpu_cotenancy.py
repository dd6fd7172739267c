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


@pytest.mark.parametrize("mode,n", [("fwd", 3000), ("step", 60), ("task", 25), ("emd", 400), ("scan", 800)])
def test_results_do_not_depend_on_a_second_process_on_the_gpu(mode, n):
    """STRICT: one deviating pass fails the test.  A build that is affected deviates in ~1 % of the passes (dozens of events in a run
    of the `fwd` mode).  The summary lines -- with the first deviation's description when there is one -- are appended to
    gpurun_out/cotenancy_first_deviation.txt (merged back from the GPU box), so that an event leaves more than a red test behind.
    `emd` covers the one object that carries packed fp32 instructions (hand-written, destinations disjoint from their sources:
    emd.hip); SAMPLENET_AMD_EMD_SCALAR=1 at build time compiles that unit without them (build.py).  `scan`: the pair scan's
    large-batch launch shape."""
    ok, lines = _stress(mode, n)
    if not ok:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "cotenancy_first_deviation.txt"), "a") as f:
                f.write("\n".join(lines) + "\n")
        except OSError:
            pass
    assert ok, "\n".join(lines)


@pytest.mark.parametrize("ncus,lds_kb", [(4, 32), (16, 64)])
def test_fc_chains_beside_resident_workgroups_on_their_xcd(tmp_path, ncus, lds_kb):
    """VERDICT r5 #7c: at N > 1 a collective library's kernels are resident beside the step's.  The FC chain launches need 8 (forward)
    and 16 (backward) workgroups of 137 KB LDS co-resident on ONE XCD; here `ncus` workgroups of a spinning stand-in kernel
    (tools/micro/resident_spin.hip, built on the spot) hold that many CUs of XCD 0 -- with 64 KB of LDS each a chain workgroup
    cannot share their CU, so 16 of them leave exactly the 16 CUs the backward chain needs -- while the captured training step
    replays on another stream: no seam times out (error words clear), no NaN, and the gradients are bit-identical to a run
    without the co-tenant."""
    import ctypes
    import subprocess

    import torch

    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    so = str(tmp_path / "libresident_spin.so")
    src = os.path.join(ROOT, "tools", "micro", "resident_spin.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src], timeout=600)
    spin = ctypes.CDLL(so)
    spin.resident_spin_start.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

    torch.manual_seed(0)
    net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    x = torch.rand(32, 1024, 3, device="cuda") - 0.5
    red = FlatGradAllReducer(net)
    step = SamplerTrainStep(net, x, reducer=red, input_ring=[x])
    step.replay(0)
    torch.cuda.synchronize()
    alone = red.flat.clone()
    loss_alone = float(step.loss)

    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    arrived = torch.zeros(1, device="cuda", dtype=torch.int32)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    try:
        assert spin.resident_spin_start(ncus, lds_kb * 1024, flag.data_ptr(), arrived.data_ptr(), 4000, side.cuda_stream) == 0
        # wait (on the host) until the stand-ins are resident: a second stream reads the counter
        probe = torch.cuda.Stream()
        with torch.cuda.stream(probe):
            for _ in range(2000):
                if int(arrived.item()) >= ncus:
                    break
        assert int(arrived.item()) == ncus, "the stand-in workgroups did not all become resident"
        done = torch.cuda.Event()
        for _ in range(20):
            step.replay(0)
        done.record()
        done.synchronize()  # (the main stream only: the stand-ins are still spinning)
        assert int(arrived.item()) == ncus and int(flag.item()) == 0
        step.check()  # raises if a chain seam timed out
        assert torch.isfinite(step.loss).item() and float(step.loss) == loss_alone
        assert torch.equal(red.flat, alone)
    finally:
        with torch.cuda.stream(torch.cuda.Stream()):
            flag.fill_(1)  # release the stand-ins
        torch.cuda.synchronize()
