"""Row f4 loaders (samplenet_amd/data.py) against fixtures written / computed by the reference itself
(tests/golden/make_golden.py golden_loaders: the vendored plyfile package and in_out.py's own split_data / PointCloudDataSet).
CPU only."""
import os

import numpy as np
import pytest

from samplenet_amd import data as D

PLY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ply")
CASES = [("02691156", "a1"), ("02691156", "b2"), ("03001627", "c3"), ("03001627", "d4"), ("04379243", "e5")]


@pytest.mark.parametrize("syn,model", CASES)
def test_load_ply_matches_reference_reader(golden, syn, model):
    g = golden("loaders_reference.npz")
    path = os.path.join(PLY, syn, model + ".ply")
    pts = D.load_ply(path)
    assert pts.dtype == np.float32 and np.array_equal(pts, g[f"{syn}_{model}_points"])
    pts2, faces, color = D.load_ply(path, with_faces=True, with_color=True)
    assert np.array_equal(pts2, pts)
    assert np.array_equal(faces, g[f"{syn}_{model}_faces"]) and np.array_equal(color, g[f"{syn}_{model}_color"])
    p, mid, sid = D.pc_loader(path)
    assert (mid, sid) == (model, syn) and np.array_equal(p, pts)


def test_folder_loading_and_split(golden):
    ds = D.load_all_point_clouds_under_folder(PLY, n_threads=3, file_ending=".ply")
    assert ds.num_examples == 5 and ds.n_points == 12 and ds.point_clouds.dtype == np.float32
    assert sorted(ds.labels.tolist()) == sorted("%s_%s" % c for c in CASES)
    g = golden("loaders_reference.npz")
    for lab, pc in zip(ds.labels, ds.point_clouds):
        syn, model = lab.split("_")
        assert np.array_equal(pc, g[f"{syn}_{model}_points"])
    tr, va, te = D.load_and_split_all_point_clouds_under_folder(PLY, n_threads=2, split=(0.6, 0.2, 0.2), seed=42)
    assert (tr.num_examples, va.num_examples, te.num_examples) == (3, 1, 1)
    assert sorted(np.concatenate([tr.labels, va.labels, te.labels]).tolist()) == sorted(ds.labels.tolist())


def test_split_data_and_dataset_iteration_match_reference(golden):
    g = golden("loaders_reference.npz")
    data = g["split_data"]
    tr, va, te, perm = D.split_data(data, (0.85, 0.05, 0.10), 42)
    assert np.array_equal(perm, g["split_perm"])
    assert np.array_equal(tr, g["split_train"]) and np.array_equal(va, g["split_val"]) and np.array_equal(te, g["split_test"])
    labels = np.array(["m%d" % i for i in range(23)], dtype=object)
    ds = D.PointCloudDataSet(data, labels=labels, init_shuffle=False)
    np.random.seed(7)
    ds.shuffle_points(seed=3)
    seq = [ds.next_batch(8, seed=11)[0].copy() for _ in range(7)]
    assert np.array_equal(np.stack(seq), g["ds_batches"]) and ds.epochs_completed == int(g["ds_epochs"])
    fe, fl, ns = ds.full_epoch_data(shuffle=True, seed=13)
    assert ns is None and np.array_equal(fe, g["ds_full_epoch"]) and [str(v) for v in fl] == g["ds_full_labels"].tolist()
    other = D.PointCloudDataSet(data[:4], labels=labels[:4], init_shuffle=False)
    assert ds.merge(other).num_examples == 27


def test_ply_list_and_scalar_corner_cases(tmp_path):
    # ragged list property + double / short scalars, ascii and both binary byte orders, comments in the header
    import struct

    hdr = ("ply\nformat %s 1.0\ncomment made by hand\nelement vertex 2\nproperty double x\nproperty float y\nproperty short z\n"
           "element poly 2\nproperty list uchar uint idx\nend_header\n")
    verts = [(0.5, 1.5, -3), (2.25, -1.0, 7)]
    polys = [[1, 2, 3], [4, 5, 6, 7]]
    p = tmp_path / "a.ply"
    p.write_text(hdr % "ascii" + "".join("%r %r %d\n" % v for v in verts) + "".join("%d %s\n" % (len(q), " ".join(map(str, q))) for q in polys))
    for name, e in (("le.ply", "<"), ("be.ply", ">")):
        body = b"".join(struct.pack(e + "dfh", *v) for v in verts)
        body += b"".join(struct.pack(e + "B%dI" % len(q), len(q), *q) for q in polys)
        (tmp_path / name).write_bytes((hdr % ("binary_little_endian" if e == "<" else "binary_big_endian")).encode() + body)
    for name in ("a.ply", "le.ply", "be.ply"):
        ply = D.read_ply(str(tmp_path / name))
        assert ply["vertex"]["x"].dtype == np.float64 and ply["vertex"]["z"].dtype == np.int16
        assert ply["vertex"]["x"].tolist() == [0.5, 2.25] and ply["vertex"]["y"].tolist() == [1.5, -1.0]
        assert ply["vertex"]["z"].tolist() == [-3, 7]
        assert [r.tolist() for r in ply["poly"]["idx"]] == polys
    with pytest.raises(ValueError):
        (tmp_path / "bad.ply").write_text("plyx\n")
        D.read_ply(str(tmp_path / "bad.ply"))


def test_modelnet_loader_needs_its_shards(tmp_path):
    with pytest.raises(FileNotFoundError):
        D.ModelNetCls(1024, None, train=True, base_dir=str(tmp_path))


def test_device_batch_ring_on_cpu():
    import torch

    ring = D.DeviceBatchRing(4, 16, "cpu", depth=2)
    a = np.random.default_rng(0).random((4, 16, 3), dtype=np.float32)
    ring.load(1, a)
    assert torch.equal(ring.ready(1), torch.from_numpy(a)) and len(ring) == 2
    ring.release(1)
    with pytest.raises(ValueError):
        ring.load(0, a[:2])


@pytest.mark.parametrize("train", [True, False])
def test_modelnet_loader_matches_reference(golden, train):
    """ModelNetCls against the REFERENCE class run on the same shard set (tests/golden/make_golden.py golden_modelnet: the
    reference module imported with a stand-in h5py that opens the .npz twins of the shards).  Here the shards are read
    through shard_reader= (h5py is not installed): everything but the h5py call itself is exercised -- file lists,
    concatenation, label shape, per-item point order from numpy's global generator, shape names."""
    g = golden("modelnet_reference.npz")
    split = "train" if train else "test"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modelnet")

    def npz(path):
        with np.load(path) as f:
            return f["data"], f["label"]

    ds = D.ModelNetCls(24, None, train=train, folder="modelnet40_ply_hdf5_2048", include_shapes=True, base_dir=root, shard_reader=npz)
    assert len(ds) == g[split + "_points"].shape[0] and ds.num_points == 24
    assert np.array_equal(ds.points, g[split + "_all_points"]) and np.array_equal(ds.labels, g[split + "_all_labels"])
    np.random.seed(5)
    for i in range(len(ds)):
        pts, lab, shape = ds[i]
        assert np.array_equal(pts, g[split + "_points"][i]) and lab.dtype == __import__("torch").int64
        assert np.array_equal(lab.numpy(), g[split + "_labels"][i]) and shape == str(g[split + "_shapes"][i])
    ds.set_num_points(10 ** 6)
    assert ds.num_points == 40
    scaled = D.ModelNetCls(8, lambda c: c * 2, train=train, folder="modelnet40_ply_hdf5_2048", base_dir=root, shard_reader=npz)
    np.random.seed(5)
    pts, lab = scaled[0]
    assert pts.shape == (8, 3) and np.array_equal(pts, 2 * ds.points[0][np.random.RandomState(5).permutation(8)])


def test_modelnet_loader_without_h5py_says_so(tmp_path):
    import importlib.util

    if importlib.util.find_spec("h5py") is not None:
        pytest.skip("h5py is installed here")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modelnet")
    with pytest.raises(ImportError, match="shard_reader"):
        D.ModelNetCls(24, None, train=True, folder="modelnet40_ply_hdf5_2048", base_dir=root)
