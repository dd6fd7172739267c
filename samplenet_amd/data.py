"""Data side of the hot path: the loaders either side of the sampler step.

  * PLY point clouds + the ShapeNet folder scheme      -- reconstruction/src/in_out.py:134-403 (load_ply, pc_loader,
    load_all_point_clouds_under_folder, load_and_split_..., split_data, PointCloudDataSet), same names / arguments / return
    conventions / np.random call sequence, so the reconstruction scripts' splits and epoch order are reproduced.  The PLY
    parser is written here against the format itself (header grammar, ascii / binary_little_endian / binary_big_endian
    bodies, scalar and list properties) instead of the vendored plyfile package.
  * ModelNet40 HDF5 shards                              -- registration/data/modelnet_loader_torch.py:20-127 (ModelNetCls);
    needs h5py like the reference (absent in the build image: the class raises ImportError on construction there).
  * DeviceBatchRing                                     -- new: the MI355X-side hand-over.  Batches are staged in pinned host
    memory and copied asynchronously (their own HIP stream) into the resident device tensors that engine.SamplerTrainStep's
    captured graphs read (input_ring): the copy of batch i+1 overlaps the step on batch i, nothing is allocated per step.
"""
import json
import os
import os.path as osp
import re
import warnings

import numpy as np

# ----------------------------------------------------------------------------------------------------- PLY
_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def _parse_ply_header(f):
    if f.readline().strip() != b"ply":
        raise ValueError("not a PLY file")
    fmt, elements = None, []
    while True:
        line = f.readline()
        if not line:
            raise ValueError("PLY header is not terminated by end_header")
        tok = line.decode("ascii", "replace").split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
        elif tok[0] == "property":
            if not elements:
                raise ValueError("PLY property outside an element")
            if tok[1] == "list":
                elements[-1]["props"].append((tok[4], ("list", _PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]])))
            else:
                elements[-1]["props"].append((tok[2], _PLY_TYPES[tok[1]]))
        elif tok[0] == "end_header":
            break
        else:
            raise ValueError("unknown PLY header keyword %r" % tok[0])
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError("unsupported PLY format %r" % fmt)
    return fmt, elements


def read_ply(file_name):
    """-> {element name: {property name: array}} ; scalar properties as 1-D arrays of the file's type, list properties as
    object arrays of 1-D arrays (a 2-D array when every row has the same length, e.g. triangle faces)."""
    with open(file_name, "rb") as f:
        fmt, elements = _parse_ply_header(f)
        order = {"ascii": "=", "binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        out = {}
        for el in elements:
            n, props = el["count"], el["props"]
            scalar = all(not isinstance(t, tuple) for _, t in props)
            if fmt != "ascii" and scalar:
                dt = np.dtype([(name, order + t) for name, t in props])
                rec = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
                out[el["name"]] = {name: np.ascontiguousarray(rec[name]).astype(t) for name, t in props}
                continue
            cols = {name: [] for name, _ in props}
            for _ in range(n):
                if fmt == "ascii":
                    tok = f.readline().split()
                    pos = 0
                    for name, t in props:
                        if isinstance(t, tuple):
                            k = int(tok[pos])
                            cols[name].append(np.array(tok[pos + 1:pos + 1 + k], dtype=np.float64).astype(t[2]))
                            pos += 1 + k
                        else:
                            cols[name].append(np.float64(tok[pos]))
                            pos += 1
                else:
                    for name, t in props:
                        if isinstance(t, tuple):
                            cdt, vdt = np.dtype(order + t[1]), np.dtype(order + t[2])
                            k = int(np.frombuffer(f.read(cdt.itemsize), dtype=cdt)[0])
                            cols[name].append(np.frombuffer(f.read(vdt.itemsize * k), dtype=vdt).astype(t[2]))
                        else:
                            sdt = np.dtype(order + t)
                            cols[name].append(np.frombuffer(f.read(sdt.itemsize), dtype=sdt)[0])
            conv = {}
            for name, t in props:
                if isinstance(t, tuple):
                    rows = cols[name]
                    if rows and all(len(r) == len(rows[0]) for r in rows):
                        conv[name] = np.vstack(rows) if rows else np.zeros((0, 0), dtype=t[2])
                    else:
                        arr = np.empty(len(rows), dtype=object)
                        arr[:] = rows
                        conv[name] = arr
                else:
                    conv[name] = np.asarray(cols[name]).astype(t)
            out[el["name"]] = conv
    return out


def load_ply(file_name, with_faces=False, with_color=False):
    """in_out.py:143-163: points (n,3) [, faces (f,3)] [, colors (n,3)]; a single array when nothing else is requested."""
    ply = read_ply(file_name)
    v = ply["vertex"]
    points = np.vstack([v["x"], v["y"], v["z"]]).T
    ret_val = [points]
    if with_faces:
        ret_val.append(np.vstack(ply["face"]["vertex_indices"]))
    if with_color:
        ret_val.append(np.hstack((np.vstack(v["red"]), np.vstack(v["green"]), np.vstack(v["blue"]))))
    return ret_val[0] if len(ret_val) == 1 else ret_val


def files_in_subdirs(top_dir, search_pattern):
    regex = re.compile(search_pattern)
    for path, _, files in os.walk(top_dir):
        for name in files:
            full_name = osp.join(path, name)
            if regex.search(full_name):
                yield full_name


def pc_loader(f_name):
    """Point cloud saved under ShapeNet's folder scheme /syn_id/model_name.ply -> (points, model_id, syn_id)  (in_out.py:166-173)."""
    tokens = f_name.split("/")
    return load_ply(f_name), tokens[-1].split(".")[0], tokens[-2]


def load_point_clouds_from_filenames(file_names, n_threads, loader, verbose=False):
    """in_out.py:220-243 (the reference fans the files out over a multiprocessing.Pool; a thread pool gives the same order
    without forking a process that holds a GPU context)."""
    from concurrent.futures import ThreadPoolExecutor

    pc = loader(file_names[0])[0]
    pclouds = np.empty([len(file_names), pc.shape[0], pc.shape[1]], dtype=np.float32)
    model_names = np.empty([len(file_names)], dtype=object)
    class_ids = np.empty([len(file_names)], dtype=object)
    with ThreadPoolExecutor(max_workers=max(1, int(n_threads))) as pool:
        for i, data in enumerate(pool.map(loader, file_names)):
            pclouds[i, :, :], model_names[i], class_ids[i] = data
    if len(np.unique(model_names)) != len(pclouds):
        warnings.warn("Point clouds with the same model name were loaded.")
    if verbose:
        print("{0} pclouds were loaded. They belong in {1} shape-classes.".format(len(pclouds), len(np.unique(class_ids))))
    return pclouds, model_names, class_ids


def split_data(data, split, seed, perm=None):
    """in_out.py:246-275: the same np.random call sequence, hence the same permutation for a seed."""
    assert sum(split) == 1.0, "data split does not sum to 1: %.2f" % sum(split)
    num_examples = data.shape[0]
    if perm is not None:
        assert perm.shape[0] == data.shape[0], "perm.shape: %s data.shape: %s" % (perm.shape, data.shape)
    else:
        if seed is not None:
            np.random.seed(seed)
        perm = np.arange(num_examples)
        np.random.shuffle(perm)
    data = data[perm]
    train_end = int(round(split[0] * num_examples))
    val_end = int(round((split[0] + split[1]) * num_examples))
    train_data, val_data, test_data = data[:train_end], data[train_end:val_end], data[val_end:]
    counts = [train_data.shape[0], val_data.shape[0], test_data.shape[0]]
    assert sum(counts) == num_examples, "data split (%d, %d, %d) does not sum to num_examples (%d)" % (*counts, num_examples)
    return train_data, val_data, test_data, perm


class PointCloudDataSet(object):
    """in_out.py:278-403 (MNIST-style epoch iterator over an (n, points, 3) array; labels; optional noisy copies)."""

    def __init__(self, point_clouds, noise=None, labels=None, copy=True, init_shuffle=True):
        self.num_examples = point_clouds.shape[0]
        self.n_points = point_clouds.shape[1]
        if labels is not None:
            assert point_clouds.shape[0] == labels.shape[0], "points.shape: %s labels.shape: %s" % (point_clouds.shape, labels.shape)
            self.labels = labels.copy() if copy else labels
        else:
            self.labels = np.ones(self.num_examples, dtype=np.int8)
        if noise is not None:
            assert type(noise) is np.ndarray
            self.noisy_point_clouds = noise.copy() if copy else noise
        else:
            self.noisy_point_clouds = None
        self.point_clouds = point_clouds.copy() if copy else point_clouds
        self.epochs_completed = 0
        self._index_in_epoch = 0
        if init_shuffle:
            self.shuffle_data()

    def shuffle_data(self, seed=None):
        if seed is not None:
            np.random.seed(seed)
        perm = np.arange(self.num_examples)
        np.random.shuffle(perm)
        self.point_clouds = self.point_clouds[perm]
        self.labels = self.labels[perm]
        if self.noisy_point_clouds is not None:
            self.noisy_point_clouds = self.noisy_point_clouds[perm]
        return self

    def shuffle_points(self, seed=None):
        if seed is not None:
            np.random.seed(seed)
        perm = np.arange(self.n_points)
        for i in range(self.num_examples):
            np.random.shuffle(perm)
            self.point_clouds[i, :, :] = self.point_clouds[i, perm, :]
            if self.noisy_point_clouds is not None:
                self.noisy_point_clouds[i, :, :] = self.noisy_point_clouds[i, perm, :]
        return self

    def next_batch(self, batch_size, seed=None):
        """The next batch_size examples: (point_clouds, labels, noisy_point_clouds | None)."""
        start = self._index_in_epoch
        self._index_in_epoch += batch_size
        if self._index_in_epoch > self.num_examples:
            self.epochs_completed += 1
            self.shuffle_data(seed)
            start = 0
            self._index_in_epoch = batch_size
        end = self._index_in_epoch
        noisy = None if self.noisy_point_clouds is None else self.noisy_point_clouds[start:end]
        return self.point_clouds[start:end], self.labels[start:end], noisy

    def full_epoch_data(self, shuffle=True, seed=None):
        if shuffle and seed is not None:
            np.random.seed(seed)
        perm = np.arange(self.num_examples)
        if shuffle:
            np.random.shuffle(perm)
        ns = None if self.noisy_point_clouds is None else self.noisy_point_clouds[perm]
        return self.point_clouds[perm], self.labels[perm], ns

    def merge(self, other_data_set):
        self._index_in_epoch = 0
        self.epochs_completed = 0
        self.point_clouds = np.vstack((self.point_clouds, other_data_set.point_clouds))
        labels_1 = self.labels.reshape([self.num_examples, 1])
        labels_2 = other_data_set.labels.reshape([other_data_set.num_examples, 1])
        self.labels = np.squeeze(np.vstack((labels_1, labels_2)))
        if self.noisy_point_clouds is not None:
            self.noisy_point_clouds = np.vstack((self.noisy_point_clouds, other_data_set.noisy_point_clouds))
        self.num_examples = self.point_clouds.shape[0]
        return self


def load_all_point_clouds_under_folder(top_dir, n_threads=20, file_ending=".ply", verbose=False):
    file_names = [f for f in files_in_subdirs(top_dir, file_ending)]
    pclouds, model_ids, syn_ids = load_point_clouds_from_filenames(file_names, n_threads, loader=pc_loader, verbose=verbose)
    return PointCloudDataSet(pclouds, labels=syn_ids + "_" + model_ids, init_shuffle=False)


def load_and_split_all_point_clouds_under_folder(top_dir, n_threads=20, file_ending=".ply", split=(0.85, 0.05, 0.10), seed=42,
                                                 verbose=False):
    file_names = [f for f in files_in_subdirs(top_dir, file_ending)]
    pclouds, model_ids, syn_ids = load_point_clouds_from_filenames(file_names, n_threads, loader=pc_loader, verbose=verbose)
    pc_tr, pc_va, pc_te, perm = split_data(pclouds, split, seed)
    mi_tr, mi_va, mi_te, _ = split_data(model_ids, split, seed, perm)
    si_tr, si_va, si_te, _ = split_data(syn_ids, split, seed, perm)
    return (PointCloudDataSet(pc_tr, labels=si_tr + "_" + mi_tr, init_shuffle=False),
            PointCloudDataSet(pc_va, labels=si_va + "_" + mi_va, init_shuffle=False),
            PointCloudDataSet(pc_te, labels=si_te + "_" + mi_te, init_shuffle=False))


# ----------------------------------------------------------------------------------------------------- ModelNet40 (HDF5)
def _get_data_files(list_filename):
    with open(list_filename) as f:
        return [line.rstrip()[5:] for line in f]


def _load_data_file(name):
    import h5py  # like the reference; not part of the build image (ModelNetCls raises ImportError there)

    with h5py.File(name, "r") as f:
        return f["data"][:], f["label"][:]


class ModelNetCls(object):
    """registration/data/modelnet_loader_torch.py:32-127: ModelNet40 point clouds from the `modelnet40_ply_hdf5_2048` shards.
    A map-style dataset (__getitem__ / __len__: usable with torch.utils.data.DataLoader as the reference's is).  The shards
    must already be on disk under `base_dir` (the reference downloads them with curl; there is no network here:
    download=True raises)."""

    def __init__(self, num_points, transforms, train, download=False, cinfo=None, folder="modelnet10_hdf5_2048", url=None,
                 include_shapes=False, base_dir=None):
        self.transforms = transforms
        self.folder = folder
        base_dir = base_dir or os.getcwd()
        self.data_dir = os.path.join(base_dir, self.folder)
        if not os.path.exists(self.data_dir):
            raise FileNotFoundError("ModelNet shards not found under %s%s" % (
                self.data_dir, " (downloading is not supported: fetch %s there)" % url if download else ""))
        self.train = train
        self.files = _get_data_files(os.path.join(self.data_dir, "train_files.txt" if train else "test_files.txt"))
        point_list, label_list = [], []
        for f in self.files:
            points, labels = _load_data_file(os.path.join(base_dir, f))
            point_list.append(points)
            label_list.append(labels)
        self.points = np.concatenate(point_list, 0)
        self.labels = np.concatenate(label_list, 0)
        if np.ndim(self.labels) == 1:
            self.labels = np.expand_dims(self.labels, axis=1)
        self.set_num_points(num_points)
        self.classes, self.class_to_idx = cinfo if cinfo is not None else (None, None)
        self.shapes = []
        self.include_shapes = include_shapes
        if self.include_shapes:
            T = "train" if self.train else "test"
            for n in range(len(self.files)):
                with open(os.path.join(self.data_dir, "ply_data_%s_%d_id2file.json" % (T, n)), "r") as f:
                    self.shapes += json.load(f)

    def __getitem__(self, idx):
        import torch

        pt_idxs = np.arange(0, self.num_points)
        np.random.shuffle(pt_idxs)
        current_points = self.points[idx, pt_idxs].copy()
        label = torch.from_numpy(self.labels[idx]).type(torch.LongTensor)
        if self.transforms is not None:
            current_points = self.transforms(current_points)
        if self.include_shapes:
            return current_points, label, self.shapes[idx]
        return current_points, label

    def __len__(self):
        return self.points.shape[0]

    def set_num_points(self, pts):
        self.num_points = min(self.points.shape[1], pts)

    def randomize(self):
        pass


# ----------------------------------------------------------------------------------------------------- host -> HBM hand-over
class DeviceBatchRing(object):
    """Resident device batches for engine.SamplerTrainStep(input_ring=ring.tensors): `depth` tensors of shape
    (batch, n_points, 3) on `device`, each with a pinned host staging buffer.  load(i, clouds) copies a host batch
    (numpy (batch, n, 3) or a CPU tensor) into slot i on the ring's own copy stream; ready(i) makes the CURRENT stream wait
    for that copy -- call it right before step.replay(i).  Typical loop (copy of batch t+1 overlaps the step on batch t):

        ring.load(0, ds.next_batch(B)[0])
        for t in range(steps):
            i = t % depth
            ring.load((i + 1) % depth, ds.next_batch(B)[0])
            ring.ready(i)
            step.replay(i)
    """

    def __init__(self, batch, n_points, device, depth=2):
        import torch

        if depth < 2:
            raise ValueError("a ring needs at least two slots")
        self.device = torch.device(device)
        self.tensors = [torch.empty(batch, n_points, 3, device=self.device, dtype=torch.float32) for _ in range(depth)]
        pin = self.device.type == "cuda"
        self._host = [torch.empty(batch, n_points, 3, dtype=torch.float32, pin_memory=pin) for _ in range(depth)]
        self._stream = torch.cuda.Stream(device=self.device) if pin else None
        self._done = [None] * depth
        self._consumed = [None] * depth

    def __len__(self):
        return len(self.tensors)

    def load(self, i, clouds):
        import torch

        src = torch.from_numpy(np.ascontiguousarray(clouds, dtype=np.float32)) if isinstance(clouds, np.ndarray) else clouds
        if tuple(src.shape) != tuple(self.tensors[i].shape):
            raise ValueError("batch of shape %s does not fit ring slot %s" % (tuple(src.shape), tuple(self.tensors[i].shape)))
        if self._stream is None:
            self.tensors[i].copy_(src)
            return
        if self._done[i] is not None:
            self._done[i].synchronize()  # the staging buffer of this slot is free again
        self._host[i].copy_(src)
        if self._consumed[i] is not None:
            self._stream.wait_event(self._consumed[i])  # the step that read this slot has finished
        with torch.cuda.stream(self._stream):
            self.tensors[i].copy_(self._host[i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._done[i] = ev

    def ready(self, i):
        """Order the current stream behind slot i's copy; returns the slot's tensor."""
        import torch

        if self._stream is not None and self._done[i] is not None:
            torch.cuda.current_stream(self.device).wait_event(self._done[i])
        return self.tensors[i]

    def release(self, i):
        """Call after enqueueing the work that reads slot i (step.replay(i)): the next load(i) waits for it."""
        import torch

        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._consumed[i] = ev
