"""CPU-side tests (no GPU, no compute calls): the C-ABI library builds, loads, and exports every symbol that
include/samplenet_hip.h declares; the ctypes prototype table covers the header; host-side utilities behave like the
reference's (sputils); the product refuses to run without a GPU instead of falling back."""
import ctypes
import importlib.util
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    path = os.path.join(ROOT, "samplenet_amd", "lib", "libsamplenet_hip.so")
    if not os.path.exists(path):  # hipcc cross-compiles gfx950 without a GPU
        spec = importlib.util.spec_from_file_location("sn_build", os.path.join(ROOT, "samplenet_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    return path


def _header_functions(which=("samplenet_hip.h", "samplenet_hip_internal.h")):
    names = set()
    for h in which:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol(libpath):
    import torch  # noqa: F401  (its bundled HIP runtime satisfies the library's libamdhip64 dependency)

    lib = ctypes.CDLL(libpath)
    names = _header_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.sn_abi_version.restype = ctypes.c_int
    assert lib.sn_abi_version() == 1
    lib.sn_workspace_bytes.restype = ctypes.c_longlong
    lib.sn_workspace_bytes.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 4
    # reference op allocates (b,(n+m)*2) floats (tf_approxmatch.cpp:167-168); ours keeps all 10 levels' ratio vectors
    assert lib.sn_workspace_bytes(b"approxmatch", 2, 100, 50, 0) == 2 * (100 + 50) * 11 * 4
    assert lib.sn_workspace_bytes(b"matchcost", 3, 600, 50, 0) == 3 * 3 * 4


def test_public_header_is_the_drop_in_boundary():
    """include/samplenet_hip.h holds the entries that replace a reference interface and nothing of the fused-step
    plumbing; samplenet_hip_internal.h holds the rest."""
    public = set(_header_functions(("samplenet_hip.h",)))
    internal = set(_header_functions(("samplenet_hip_internal.h",)))
    assert not public & internal
    for name in ("sn_pairscan_forward", "sn_knn", "sn_chamfer_forward", "sn_chamfer_backward", "sn_group_point", "sn_group_point_grad",
                 "sn_grouping_operation", "sn_soft_project_backward", "sn_approxmatch", "sn_matchcost", "sn_matchcost_grad",
                 "sn_linear_forward", "sn_linear_dgrad", "sn_linear_wgrad", "sn_abi_version", "sn_last_error_string"):
        assert name in public, name  # SURVEY 8b's list of what a C-ABI replacement must export
    assert not [n for n in public if re.search(r"step|fc_chain|conv_stack|_keys|_partial|tail", n)], public
    assert len(public) <= 48  # (round 4: + the three *_backward_ordered forms -- same reference interfaces, deterministic sums)


def test_python_prototypes_cover_the_header(libpath):
    from samplenet_amd import _lib

    assert set(_header_functions()) == set(_lib.PROTOTYPES), set(_header_functions()) ^ set(_lib.PROTOTYPES)


def test_argument_errors_are_reported_without_a_gpu(libpath):
    """Argument validation happens before any device work: error codes + messages work on a GPU-less host."""
    from samplenet_amd._lib import SampleNetHipError, check, lib

    rc = lib.sn_chamfer_forward(2, -1, None, 4, None, None, None, None, None, None)
    assert rc == 10001 and b"negative" in lib.sn_last_error_string()
    with pytest.raises(SampleNetHipError):
        check(lib.sn_knn(1, 10, 4, 0, None, 0, None, 0, None, None, None), "sn_knn")
    assert lib.sn_pairscan_forward(1, 4, 2, 8, ctypes.c_void_p(8), 0, ctypes.c_void_p(8), 0, None, None, None, None, None,
                                   None, None, 0, None, None, 0.0, None) == 10001  # K > N
    assert lib.sn_chamfer_forward(0, 5, None, 7, None, None, None, None, None, None) == 0  # empty batch: no-op


def test_no_cpu_fallback():
    import samplenet_amd
    from samplenet_amd import ChamferDistance, SampleNet, SoftProjection

    with pytest.raises(RuntimeError):
        ChamferDistance()(torch.zeros(1, 4, 3), torch.zeros(1, 5, 3))
    with pytest.raises(RuntimeError):
        SoftProjection(2)(torch.zeros(1, 3, 8), torch.zeros(1, 3, 4))
    net = SampleNet(4, 8, 2, input_shape="bnc", output_shape="bnc")
    with pytest.raises(RuntimeError):
        net(torch.zeros(2, 16, 3))
    assert not any(name.startswith("oracle") for name in list(__import__("sys").modules) if "samplenet_amd" in name)


def test_state_dict_keys_match_reference(golden):
    from samplenet_amd import SampleNet

    g = golden("samplenet_reference.npz")
    ref_keys = sorted(k[len("c1_sd_"):] for k in g.files if k.startswith("c1_sd_"))
    net = SampleNet(64, 128, 8, input_shape="bnc", output_shape="bnc")
    assert sorted(net.state_dict().keys()) == ref_keys
    assert net.name == "samplenet"
    with pytest.raises(ValueError):
        SampleNet(8, 16, 4, output_shape="cbn")


def test_sputils_matches_reference_golden(golden):
    from samplenet_amd import sputils

    g = golden("nn_matching_reference.npz")
    k = g["idx"].shape[1]
    assert np.array_equal(sputils.nn_matching(g["pc"], g["idx"], k, complete_fps=True), g["out_fps"])
    assert np.array_equal(sputils.nn_matching(g["pc"], g["idx"], k, complete_fps=False), g["out_nofps"])
    a = sputils.get_parser().parse_args([])
    assert (a.num_in_points, a.num_out_points, a.bottleneck_size, a.projection_group_size) == (1024, 64, 128, 8)
    assert (a.alpha, a.gamma, a.delta, a.lmbda, a.skip_projection) == (0.01, 1, 0, 0.01, False)


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus N` outside torch.distributed.run (the driver's call) starts its own N ranks and passes rank 0's
    JSON line through -- exercised here with the launcher self-test (gloo, CPU; no measurement); without enough GPUs the
    real call exits with a message and code 2, not a traceback."""
    import json
    import subprocess
    import sys

    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--launcher-selftest"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"launcher_selftest": True, "n_gpus": 2, "rank_sum": 1.0}
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 2 and "exposes 0 GPU" in r.stderr and "Traceback" not in r.stderr


def test_device_code_carries_no_packed_fp32_arithmetic(tmp_path):
    """The product build switches the packed fp32 VALU ops off (samplenet_amd/build.py NO_PACKED_F32): with a second process on the
    GPU, kernels carrying the compiler's SLP-packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 returned wrong low halves in ~1 % of
    their launches (tests/test_gpu_cotenancy.py watches the effect on a GPU).  Here: every object of the library is
    disassembled and must hold none of them -- a flag lost in a build script shows up without a GPU.  One exception, narrowly:
    emd.o carries HAND-WRITTEN packed instructions (inline asm with early-clobber destinations) -- each of them must write a
    register pair disjoint from every source pair (the aliasing is the stated trigger: an instruction whose sources survive it
    gives the same result however often a restored wave replays it), and it must hold a sensible number of them (the EMD sweeps
    are built on them: a lost feature flag shows up here too)."""
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    libdir = os.path.join(ROOT, "samplenet_amd", "lib")
    objs = sorted(f for f in os.listdir(libdir) if f.endswith(".o"))
    assert len(objs) >= 8
    checked = 0
    for o in objs:
        fat = str(tmp_path / (o + ".fat"))
        r = subprocess.run([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, os.path.join(libdir, o)], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            continue  # (capi_common: host code only)
        lst = subprocess.run([llvm + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], capture_output=True, text=True).stdout
        tgt = [t for t in lst.split() if "gfx950" in t]
        assert tgt, (o, lst)
        co = str(tmp_path / (o + ".co"))
        subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + tgt[0], "--input=" + fat, "--output=" + co])
        dis = subprocess.run([llvm + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        pk = re.findall(r"\bv_pk_(?:fma|mul|add)_f32\s+([^\n]*)", dis)
        if o == "emd.o":
            assert len(pk) >= 100, "emd.o: the hand-written packed sweeps are gone (%d packed instructions)" % len(pk)
            for ops in pk:
                regs = re.findall(r"v\[(\d+):(\d+)\]", ops)
                assert len(regs) >= 3, "emd.o: packed instruction with a non-VGPR-pair operand: " + ops  # (dst + >= 2 sources)
                (d0, d1), srcs = (int(regs[0][0]), int(regs[0][1])), [(int(a), int(b)) for a, b in regs[1:]]
                assert all(s1 < d0 or s0 > d1 for s0, s1 in srcs), "emd.o: destination pair aliases a source: v_pk_* " + ops
        else:
            assert len(pk) == 0, "%s carries %d packed fp32 instructions" % (o, len(pk))
        assert "s_endpgm" in dis
        checked += 1
    assert checked >= 7


def test_emd_scalar_build_switch_compiles_without_packed_instructions(tmp_path):
    """SAMPLENET_AMD_EMD_SCALAR=1 (build.py): emd.hip compiled like every other unit -- packed feature off, the pair helpers
    element by element (-DSN_EMD_SCALAR_F32=1).  The fallback for a platform on which the hand-written packed sweeps would ever
    misbehave under co-tenancy; nothing else would notice if it stopped compiling or kept a packed instruction."""
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    src = os.path.join(ROOT, "samplenet_amd", "csrc", "emd.hip")
    obj = str(tmp_path / "emd_scalar.o")
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "-c", src, "-o", obj, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "samplenet_amd", "csrc"), "-Wall",
           "-Wno-unused-function", "-Werror", "-ffp-contract=off", "-fno-slp-vectorize", "-DSN_EMD_SCALAR_F32=1",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    fat, co = obj + ".fat", obj + ".co"
    subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    lst = subprocess.run([llvm + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], capture_output=True, text=True).stdout
    tgt = [t for t in lst.split() if "gfx950" in t]
    subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + tgt[0], "--input=" + fat, "--output=" + co])
    dis = subprocess.run([llvm + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    assert "s_endpgm" in dis and "v_exp_f32" in dis
    assert not re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", dis)


def test_fp32_mfma_twin_of_the_conv_gemms_still_compiles(tmp_path):
    """pointnet_mlp.hip / pointnet_mlp_backward.hip carry a second implementation of the conv-stack GEMMs and of the fused conv
    backward on the fp32 MFMA (-DSN_BF16X3=0: exact fp32 products instead of six bf16 products of three-way split operands) -- the
    arithmetic reference of the split kernels (tests/test_gpu_mlp.py::test_fp32_mfma_twin_agrees_with_the_split_bf16_build runs it
    on the GPU).  It is not a product build, so nothing else would notice if it stopped compiling."""
    import subprocess

    procs = []
    for unit in ("pointnet_mlp", "pointnet_mlp_backward"):
        src = os.path.join(ROOT, "samplenet_amd", "csrc", unit + ".hip")
        cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "-c", src, "-o", str(tmp_path / (unit + "0.o")), "-O3", "-std=c++17", "-fPIC",
               "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "samplenet_amd", "csrc"),
               "-Wall", "-Wno-unused-function", "-Werror", "-DSN_BF16X3=0"]
        procs.append((unit, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for unit, p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, unit + ": " + err[-3000:]


def test_bench_profile_provenance_helpers():
    """bench.py's provenance fields: the kernel-source hash changes with the sources, the committed profile directory is found,
    its longest kernel is parsed out of the rocprofv3 summary, and a hash mismatch is flagged as stale."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench.csrc_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", sha)
    rnd, d = bench.profile_dir()
    assert re.fullmatch(r"r\d\d", rnd) and os.path.isdir(d)
    name, avg_us, calls = bench.longest_kernel_of_profile()
    assert name and avg_us > 1.0 and calls >= 1
    prov = bench.profile_provenance()
    assert prov["dir"] == "profiles/" + rnd and prov["current_src_sha16"] == sha
    if prov["src_sha16"] is not None:
        assert prov["stale"] == (prov["src_sha16"] != sha)
    assert bench.geometry_bytes_fwd(1024, 64, 8) == 24576  # SURVEY 8d's figure


def test_weight_plane_cache_semantics(monkeypatch):
    """task_features._weight_planes (host logic, no kernel): the scratch for the split weight planes is reused while the parameter
    OBJECT and its version counter are unchanged, flagged stale after an in-place update, separate per tag (transposed planes),
    never reported ready while a stream capture is under way, and dropped when another parameter object takes an old one's id."""
    import gc

    import torch

    from samplenet_amd import task_features as TF

    capturing = {"on": False}
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing["on"])
    TF._PLANES.clear()
    W = torch.nn.Parameter(torch.randn(8, 4))
    p1, r1 = TF._weight_planes(W.reshape(8, 4))
    p2, r2 = TF._weight_planes(W.reshape(8, 4))  # another view object of the same parameter
    assert not r1 and r2 and p1 is p2 and p1.numel() == 3 * 32 and p1.dtype == torch.bfloat16
    with torch.no_grad():
        W.mul_(2.0)
    p3, r3 = TF._weight_planes(W.reshape(8, 4))
    assert p3 is p1 and not r3  # same scratch, contents stale
    assert TF._weight_planes(W.reshape(8, 4))[1]
    pt, rt = TF._weight_planes(W.reshape(8, 4), tag="T")
    assert pt is not p1 and not rt
    capturing["on"] = True
    pc, rc = TF._weight_planes(W.reshape(8, 4))
    assert pc is p1 and not rc  # a capture always records the split
    capturing["on"] = False
    W2, W3 = torch.nn.Parameter(torch.randn(8, 4)), torch.nn.Parameter(torch.randn(8, 4))
    pm, rm = TF._weight_planes(W2, W3)
    assert pm.numel() == 3 * 64 and not rm and TF._weight_planes(W2, W3)[1]
    n_before = len(TF._PLANES)
    del W2, W3
    gc.collect()
    TF._weight_planes(torch.nn.Parameter(torch.randn(2, 2)))  # (a new entry sweeps those whose parameters are gone)
    assert len(TF._PLANES) <= n_before
