"""Two REAL ranks -- two processes, each with its own HIP SampleNet, FlatGradAllReducer and engine.SamplerTrainStep -- on the one
GPU of the test box (SURVEY 8e "test without a cluster").  torch.distributed runs on gloo (RCCL does not admit
two ranks on one device); everything else is the code path bench.py --gpus N runs: sharded batch, kernel-written gradient
bucket, one collective per step.  Checked:
  * fused  -- the captured fused step with per-rank BatchNorm statistics (the default of the data-parallel path): both ranks end
              with the same bucket, bit-equal to the mean of the two shards' gradients computed by ONE process; the FC chain
              kernels' hand-off error words stay clean with both processes' launches resident on the device at once;
  * fixed  -- BatchNorm on running statistics: two ranks x 32 clouds = one process x 64 clouds (every loss term is a batch mean);
  * sync   -- syncbn.convert_sync_batchnorm: training-mode statistics over both ranks' rows = one process on the whole batch.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_two(mode, tmp_path, state, x):
    torch.save({"state": state, "x": x.cpu()}, tmp_path / "input.pt")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_two_rank_worker.py"), str(r), "2", str(port), mode, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return [torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(2)]


def _fresh(seed=0):
    from samplenet_amd import SampleNet

    torch.manual_seed(seed)
    net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    with torch.no_grad():  # running statistics off their trivial initial values (the "fixed" mode normalises with them)
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.05)
                m.running_var.uniform_(0.8, 1.2)
    return net


def test_two_ranks_fused_step_per_rank_statistics(tmp_path):
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    net = _fresh()
    state = {k: v.cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(64, 1024, 3, device="cuda") - 0.5
    r0, r1 = _run_two("fused", tmp_path, state, x)
    assert torch.equal(r0["flat"], r1["flat"]) and float(r0["flat"].abs().sum()) > 0
    # the same three steps per shard in this one process
    flats, losses = [], []
    for r in range(2):
        local = _fresh()
        local.load_state_dict(state)
        red = FlatGradAllReducer(local)
        step = SamplerTrainStep(local, x[32 * r:32 * (r + 1)].contiguous(), reducer=red, use_graph=True)
        assert step._fast_path()
        for _ in range(3):
            loss = step(x[32 * r:32 * (r + 1)].contiguous())
        torch.cuda.synchronize()
        flats.append(red.flat.cpu().clone())
        losses.append(float(loss))
        for k, v in local.named_buffers():  # per-rank statistics moved exactly as in the rank's own process
            assert torch.equal(v.cpu(), (r0, r1)[r]["buffers"][k]), k
    assert [r0["loss"], r1["loss"]] == losses
    assert torch.equal(r0["flat"], (flats[0] + flats[1]) * 0.5)
    print("gloo took device tensors:", r0["native_gloo"])


@pytest.mark.parametrize("mode", ["fixed", "sync"])
def test_two_ranks_equal_one_process_on_the_whole_batch(tmp_path, mode):
    sys.path.insert(0, HERE)
    from _two_rank_worker import fixed_stats_features

    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    net = _fresh(1)
    state = {k: v.cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(64, 1024, 3, device="cuda") - 0.5
    r0, r1 = _run_two(mode, tmp_path, state, x)
    assert torch.equal(r0["flat"], r1["flat"])
    if mode == "fixed":
        fixed_stats_features(net)
    red = FlatGradAllReducer(net)
    step = SamplerTrainStep(net, x, reducer=red, use_graph=False, fused_loss=(mode != "fixed"))
    loss = step(x)
    torch.cuda.synchronize()
    whole = red.flat.cpu()
    # mean of the shards' batch means = the batch mean of every loss term (equal shards).  fixed: the same arithmetic per row,
    # only the sums over the batch are grouped differently (measured 2e-8 of the norm; bar 1e-6).  sync: the ranks merge fp32
    # per-rank moments in fp64 where the one process takes single-pass statistics over 64 rows, and the head's BatchNorms amplify
    # that (the thread-based test of tests/test_gpu_syncbn.py holds 2e-5 on the head's output, 3e-4 per gradient tensor):
    # measured 5e-4 of the bucket's norm; bar 1e-3
    err = float((r0["flat"] - whole).norm()) / float(whole.norm())
    print(mode, "two ranks vs one process: |d| / |g| = %.2e, loss %.8f vs mean of ranks %.8f" % (err, float(loss), 0.5 * (r0["loss"] + r1["loss"])))
    assert err <= (1e-6 if mode == "fixed" else 1e-3)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    if mode == "sync":  # running statistics follow the statistics of all 64 clouds on both ranks
        for k, v in net.named_buffers():
            if "running" in k:
                assert torch.allclose(r0["buffers"][k], v.cpu(), rtol=1e-4, atol=1e-6), k
                assert torch.equal(r0["buffers"][k], r1["buffers"][k]), k


def test_two_ranks_module_surface_on_captured_graphs(tmp_path):
    """the MODULE SURFACE (net(x) -> getters -> backward() -> reducer.reduce()) under data parallelism replays
    captured graphs whose backward writes the reducer's bucket; after five steps both ranks hold the same bucket, bit-equal to
    what the ENGINE's captured step leaves for the same shards (mean of the two shards' gradients)."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    net = _fresh()
    state = {k: v.cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(64, 1024, 3, device="cuda") - 0.5
    r0, r1 = _run_two("surface", tmp_path, state, x)
    assert r0["captured"][-1] and r1["captured"][-1] and r0["warnings"] == 0
    assert torch.equal(r0["flat"], r1["flat"]) and float(r0["flat"].abs().sum()) > 0
    flats = []
    for r in range(2):
        local = _fresh()
        local.load_state_dict(state)
        red = FlatGradAllReducer(local)
        # (the same launches as the surface's graphs: the task term outside the node; 3 warm-up steps + 2 = the ranks' 5 steps)
        step = SamplerTrainStep(local, x[32 * r:32 * (r + 1)].contiguous(), reducer=red, use_graph=True,
                                task_loss=lambda proj: proj.mean())
        for _ in range(2):
            step(x[32 * r:32 * (r + 1)].contiguous())
        torch.cuda.synchronize()
        flats.append(red.flat.cpu().clone())
        for k, v in local.named_buffers():
            assert torch.equal(v.cpu(), (r0, r1)[r]["buffers"][k]), k
    want = (flats[0] + flats[1]) * 0.5
    # the engine's node applies alpha inside the kernels, the script multiplies the getter's value outside (0.01 x 1 = 1 x 0.01):
    # the MLP gradients are the same arithmetic; the temperature's direct term is formed in-kernel on one side (1e-6)
    diff = (r0["flat"] - want).abs()
    print("surface vs engine bucket: max |d| %.3e of max |g| %.3e, %d of %d entries differ" %
          (float(diff.max()), float(want.abs().max()), int((diff > 0).sum()), diff.numel()))
    assert float(diff.max()) <= 1e-6 * float(want.abs().max()), float(diff.max())


def test_two_ranks_ddp_wrapped_module_is_synchronised_by_the_fallback(tmp_path):
    """torch DistributedDataParallel around the HIP module -- the captured surface would write .grad past
    DDP's reducer hooks; it must step aside (one warning per rank) so that DDP synchronises every step."""
    net = _fresh()
    state = {k: v.cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(64, 1024, 3, device="cuda") - 0.5
    r0, r1 = _run_two("ddp", tmp_path, state, x)
    assert not any(r0["captured"]) and not any(r1["captured"])
    assert r0["warnings"] == 1 and r1["warnings"] == 1
    assert torch.equal(r0["flat"], r1["flat"]) and float(r0["flat"].abs().sum()) > 0  # identical gradients on both ranks: synchronised
    assert r0["loss"] != r1["loss"]  # (different shards)
