"""numpy front-end of the CPU oracle (oracle/samplenet_oracle.c) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by samplenet_amd (the product path fails loudly without its HIP
library instead of falling back to anything here).

Every function takes / returns contiguous numpy arrays (float32 / int32) in the layout of the
reference op it restates; see the C file for the reference file:line of each algorithm.
"""
import ctypes
import glob
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)
_d = ctypes.POINTER(ctypes.c_double)
_ll = ctypes.POINTER(ctypes.c_longlong)


def build(ref=True):
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------ Chamfer / nn_distance
def chamfer_forward(xyz1, xyz2):
    """xyz1 (b,n,3), xyz2 (b,m,3) -> dist1 (b,n), idx1, dist2 (b,m), idx2."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32)
    i1 = np.empty((b, n), np.int32)
    d2 = np.empty((b, m), np.float32)
    i2 = np.empty((b, m), np.int32)
    lib().orc_chamfer_forward(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(d1, _f), _p(i1, _i), _p(d2, _f), _p(i2, _i))
    return d1, i1, d2, i2


def chamfer_backward(xyz1, xyz2, gd1, idx1, gd2, idx2):
    xyz1, xyz2, gd1, gd2 = _f32(xyz1), _f32(xyz2), _f32(gd1), _f32(gd2)
    idx1, idx2 = _i32(idx1), _i32(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    lib().orc_chamfer_backward(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(gd1, _f), _p(idx1, _i),
                               _p(gd2, _f), _p(idx2, _i), _p(g1, _f), _p(g2, _f))
    return g1, g2


# ------------------------------------------------------------------ kNN
def sqdist_matrix(xyz1, xyz2):
    """xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries -> dist (b,m,n)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    out = np.empty((b, m, n), np.float32)
    lib().orc_sqdist_matrix(b, n, m, c, _p(xyz1, _f), _p(xyz2, _f), _p(out, _f))
    return out


def selection_sort(dist, k):
    """dist (b,m,n) -> outi (b,m,n) int32, out (b,m,n); first k columns are the k smallest."""
    dist = _f32(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().orc_selection_sort(b, n, m, k, _p(dist, _f), _p(outi, _i), _p(out, _f))
    return outi, out


def knn_point_tf(k, xyz1, xyz2):
    """The in-tree kNN definition (tf_grouping.py:64-91): matrix + selection sort."""
    outi, out = selection_sort(sqdist_matrix(xyz1, xyz2), k)
    return out[:, :, :k].copy(), outi[:, :, :k].copy()


def knn(k, xyz1, xyz2):
    """(d, idx)-ordered kNN. xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> dist2 (b,m,k), idx (b,m,k)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    assert k <= n
    idx = np.empty((b, m, k), np.int32)
    dist = np.empty((b, m, k), np.float32)
    lib().orc_knn(b, n, m, k, _p(xyz1, _f), _p(xyz2, _f), _p(idx, _i), _p(dist, _f))
    return dist, idx


# ------------------------------------------------------------------ group_point
def group_point(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().orc_group_point(b, n, c, m, ns, _p(points, _f), _p(idx, _i), _p(out, _f))
    return out


def group_point_grad(points_shape, idx, grad_out):
    idx, grad_out = _i32(idx), _f32(grad_out)
    b, n, c = points_shape
    _, m, ns = idx.shape
    g = np.zeros((b, n, c), np.float32)
    lib().orc_group_point_grad(b, n, c, m, ns, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


def grouping_operation(feat, idx):
    """feat (b,c,n), idx (b,m,ns) -> (b,c,m,ns) (pointnet2 layout)."""
    feat, idx = _f32(feat), _i32(idx)
    b, c, n = feat.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), np.float32)
    lib().orc_grouping_operation(b, c, n, m, ns, _p(feat, _f), _p(idx, _i), _p(out, _f))
    return out


def grouping_operation_grad(feat_shape, idx, grad_out):
    idx, grad_out = _i32(idx), _f32(grad_out)
    b, c, n = feat_shape
    _, m, ns = idx.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_grouping_operation_grad(b, c, n, m, ns, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


# ------------------------------------------------------------------ SoftProjection
def softproj_forward(P, Q, idx, sigma, F=None):
    """P (b,3,n), Q (b,3,m), idx (b,m,k), optional F (b,cf,n) -> proj (b,3,m), prop (b,cf,m)|None, w (b,m,k)."""
    P, Q, idx = _f32(P), _f32(Q), _i32(idx)
    b, _, n = P.shape
    m = Q.shape[2]
    k = idx.shape[2]
    proj = np.empty((b, 3, m), np.float32)
    w = np.empty((b, m, k), np.float32)
    cf, prop = 0, None
    if F is not None:
        F = _f32(F)
        cf = F.shape[1]
        prop = np.empty((b, cf, m), np.float32)
    lib().orc_softproj_forward(b, n, m, k, cf, _p(P, _f), _p(Q, _f), _p(idx, _i), _p(F, _f),
                               ctypes.c_float(sigma), _p(proj, _f), _p(prop, _f), _p(w, _f))
    return proj, prop, w


def softproj_backward(P, Q, idx, sigma, grad_proj, want_grad_P=False):
    """-> grad_Q (b,3,m), grad_P (b,3,n)|None, grad_sigma (python float)."""
    P, Q, idx, grad_proj = _f32(P), _f32(Q), _i32(idx), _f32(grad_proj)
    b, _, n = P.shape
    m = Q.shape[2]
    k = idx.shape[2]
    gq = np.empty((b, 3, m), np.float32)
    gp = np.zeros((b, 3, n), np.float32) if want_grad_P else None
    gs = np.zeros(1, np.float64)
    lib().orc_softproj_backward(b, n, m, k, _p(P, _f), _p(Q, _f), _p(idx, _i), ctypes.c_float(sigma),
                                _p(grad_proj, _f), _p(gq, _f), _p(gp, _f), _p(gs, _d))
    return gq, gp, float(gs[0])


# ------------------------------------------------------------------ EMD
def approxmatch(xyz1, xyz2):
    """xyz1 (b,n,3), xyz2 (b,m,3) -> match (b,m,n)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.empty((b, m, n), np.float32)
    lib().orc_approxmatch(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(match, _f))
    return match


def matchcost(xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.empty((b,), np.float32)
    lib().orc_matchcost(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(match, _f), _p(cost, _f))
    return cost


def matchcost_grad(xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    lib().orc_matchcost_grad(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(match, _f), _p(g1, _f), _p(g2, _f))
    return g1, g2


# ------------------------------------------------------------------ inference matching
def nn_matching(full_pc, idx, k, complete_fps=True):
    """full_pc (b,n,3) float32, idx (b,k) int -> out (b,k,3) float64 (sputils.py:31-41)."""
    full_pc = _f32(full_pc)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    b, n, _ = full_pc.shape
    out = np.zeros((b, k, 3), np.float64)
    lib().orc_nn_matching(b, n, k, _p(full_pc, _f), _p(idx, _ll), int(bool(complete_fps)), _p(out, _d))
    return out


# ------------------------------------------------------------------ oracle/_ref (the reference's own CPU code)
def ref_dir():
    return os.path.join(_HERE, "_ref")


def have_ref():
    return bool(glob.glob(os.path.join(ref_dir(), "cd_ref*.so")))


_REF = {}


def ref_cd():
    """The reference's pybind11 module (chamfer_distance.cpp) as built into oracle/_ref."""
    if "cd" not in _REF:
        import torch  # noqa: F401  (libtorch must be loaded first)

        (path,) = glob.glob(os.path.join(ref_dir(), "cd_ref*.so"))
        spec = importlib.util.spec_from_file_location("cd_ref", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _REF["cd"] = mod
    return _REF["cd"]


def ref_chamfer_forward(xyz1, xyz2):
    """Runs the reference chamfer_distance_forward (chamfer_distance.cpp:90-111)."""
    import torch

    cd = ref_cd()
    x1 = torch.from_numpy(_f32(xyz1))
    x2 = torch.from_numpy(_f32(xyz2))
    b, n, _ = x1.shape
    m = x2.shape[1]
    d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
    i1, i2 = torch.zeros(b, n, dtype=torch.int), torch.zeros(b, m, dtype=torch.int)
    cd.forward(x1, x2, d1, d2, i1, i2)
    return d1.numpy(), i1.numpy(), d2.numpy(), i2.numpy()


def ref_chamfer_backward(xyz1, xyz2, gd1, idx1, gd2, idx2):
    """Runs the reference chamfer_distance_backward (chamfer_distance.cpp:114-177)."""
    import torch

    cd = ref_cd()
    x1, x2 = torch.from_numpy(_f32(xyz1)), torch.from_numpy(_f32(xyz2))
    g1, g2 = torch.zeros_like(x1), torch.zeros_like(x2)
    cd.backward(x1, x2, g1, g2, torch.from_numpy(_f32(gd1)), torch.from_numpy(_f32(gd2)),
                torch.from_numpy(_i32(idx1)), torch.from_numpy(_i32(idx2)))
    return g1.numpy(), g2.numpy()


def _ref_lib(name):
    if name not in _REF:
        _REF[name] = ctypes.CDLL(os.path.join(ref_dir(), name))
    return _REF[name]


class _quiet_stdout:
    """The reference harness functions printf their inputs; silence fd 1 around the call."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 1)

    def __exit__(self, *a):
        try:
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(self._saved, 1)
            os.close(self._null)
            os.close(self._saved)


def ref_selection_sort(dist, k):
    """Reference selection_sort_cpu (grouping/test/selection_sort.cpp:20-63)."""
    dist = _f32(dist)
    b, m, n = dist.shape
    idx = np.zeros((b, m, n), np.int32)
    val = np.zeros((b, m, n), np.float32)
    fn = _ref_lib("libselection_sort_ref.so")._Z18selection_sort_cpuiiiiPKfPiPf
    with _quiet_stdout():
        fn(b, n, m, k, _p(dist, _f), _p(idx, _i), _p(val, _f))
    return idx, val


def ref_group_point(points, idx):
    """Reference group_point_cpu (grouping/test/query_ball_point.cpp:52-66)."""
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.zeros((b, m, ns, c), np.float32)
    _ref_lib("libgroup_point_ref.so")._Z15group_point_cpuiiiiiPKfPKiPf(b, n, c, m, ns, _p(points, _f), _p(idx, _i), _p(out, _f))
    return out


def ref_group_point_grad(points_shape, idx, grad_out):
    """Reference group_point_grad_cpu (query_ball_point.cpp:70-84)."""
    idx, grad_out = _i32(idx), _f32(grad_out)
    b, n, c = points_shape
    _, m, ns = idx.shape
    g = np.zeros((b, n, c), np.float32)
    _ref_lib("libgroup_point_ref.so")._Z20group_point_grad_cpuiiiiiPKfPKiPf(b, n, c, m, ns, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


def ref_approxmatch_cpu(xyz1, xyz2):
    """Reference approxmatch_cpu (approxmatch.cpp:17-76): double precision, match laid out [k*m+l]."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.zeros((b, n, m), np.float32)
    _ref_lib("libapproxmatch_ref.so")._Z15approxmatch_cpuiiiPfS_S_(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(match, _f))
    return match


def ref_matchcost_cpu(xyz1, xyz2, match_nm):
    xyz1, xyz2, match_nm = _f32(xyz1), _f32(xyz2), _f32(match_nm)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.zeros((b,), np.float32)
    _ref_lib("libapproxmatch_ref.so")._Z13matchcost_cpuiiiPfS_S_S_(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(match_nm, _f), _p(cost, _f))
    return cost


def ref_matchcostgrad_cpu(xyz1, xyz2, match_nm):
    """Reference matchcostgrad_cpu (approxmatch.cpp:98-125): grad wrt xyz2 only."""
    xyz1, xyz2, match_nm = _f32(xyz1), _f32(xyz2), _f32(match_nm)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g2 = np.zeros((b, m, 3), np.float32)
    _ref_lib("libapproxmatch_ref.so")._Z17matchcostgrad_cpuiiiPfS_S_S_(b, n, m, _p(xyz1, _f), _p(xyz2, _f), _p(match_nm, _f), _p(g2, _f))
    return g2
