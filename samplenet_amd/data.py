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
    """Every file below top_dir whose full path matches the regular expression (in_out.py:134-140), in os.walk order."""
    matches = re.compile(search_pattern).search
    for folder, _, names in os.walk(top_dir):
        yield from filter(matches, (osp.join(folder, n) for n in names))


def pc_loader(f_name):
    """Point cloud saved under ShapeNet's folder scheme /syn_id/model_name.ply -> (points, model_id, syn_id)  (in_out.py:166-173)."""
    folder, leaf = osp.split(f_name)
    return load_ply(f_name), leaf.split(".")[0], osp.basename(folder)


def load_point_clouds_from_filenames(file_names, n_threads, loader, verbose=False):
    """-> (clouds (files, points, 3) float32, model names, class ids) in the order of file_names (in_out.py:220-243; the
    reference fans the files out over a multiprocessing.Pool -- a thread pool keeps the order without forking a process that
    holds a GPU context)."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=max(1, int(n_threads))) as pool:
        loaded = list(pool.map(loader, file_names))
    pclouds = np.stack([np.asarray(pc, dtype=np.float32) for pc, _, _ in loaded])
    model_names = np.array([m for _, m, _ in loaded], dtype=object)
    class_ids = np.array([c for _, _, c in loaded], dtype=object)
    if len(set(model_names.tolist())) != len(loaded):
        warnings.warn("Point clouds with the same model name were loaded.")
    if verbose:
        print("{0} pclouds were loaded. They belong in {1} shape-classes.".format(len(loaded), len(set(class_ids.tolist()))))
    return pclouds, model_names, class_ids


def _legacy_permutation(n, seed=None):
    """A permutation of range(n) from numpy's GLOBAL legacy generator, consuming it exactly as the reference's
    `perm = np.arange(n); np.random.shuffle(perm)` does (RandomState.permutation(int) is that pair of calls), after an
    optional re-seed: the reconstruction scripts' splits and epoch orders depend on this stream."""
    if seed is not None:
        np.random.seed(seed)
    return np.random.permutation(n)


def split_data(data, split, seed, perm=None):
    """(train, val, test, perm): rows of `data` permuted (by `perm`, or by the seeded global generator) and cut at the
    rounded cumulative fractions of `split`  (in_out.py:246-275)."""
    n = data.shape[0]
    if abs(sum(split) - 1.0) > 0:
        raise AssertionError("data split does not sum to 1: %.2f" % sum(split))
    if perm is None:
        perm = _legacy_permutation(n, seed)
    elif perm.shape[0] != n:
        raise AssertionError("perm.shape: %s data.shape: %s" % (perm.shape, data.shape))
    cuts = [int(round(split[0] * n)), int(round((split[0] + split[1]) * n))]
    train, val, test = np.split(data[perm], cuts)
    if train.shape[0] + val.shape[0] + test.shape[0] != n:
        raise AssertionError("data split (%d, %d, %d) does not sum to num_examples (%d)" % (train.shape[0], val.shape[0], test.shape[0], n))
    return train, val, test, perm


class PointCloudDataSet(object):
    """Epoch iterator over an (examples, points, 3) array with labels and optional noisy copies: the interface of
    in_out.py:278-403 (attributes point_clouds / labels / noisy_point_clouds / num_examples / n_points / epochs_completed;
    shuffle_data, shuffle_points, next_batch, full_epoch_data, merge), drawing from numpy's global generator in the same
    order so that a seeded run visits the same batches."""

    def __init__(self, point_clouds, noise=None, labels=None, copy=True, init_shuffle=True):
        own = (lambda a: a.copy()) if copy else (lambda a: a)
        self.num_examples, self.n_points = point_clouds.shape[0], point_clouds.shape[1]
        if labels is None:
            self.labels = np.ones(self.num_examples, dtype=np.int8)
        else:
            if labels.shape[0] != self.num_examples:
                raise AssertionError("points.shape: %s labels.shape: %s" % (point_clouds.shape, labels.shape))
            self.labels = own(labels)
        if noise is not None and not isinstance(noise, np.ndarray):
            raise AssertionError("noise must be a numpy array")
        self.noisy_point_clouds = None if noise is None else own(noise)
        self.point_clouds = own(point_clouds)
        self.epochs_completed, self._cursor = 0, 0
        if init_shuffle:
            self.shuffle_data()

    def _rows(self, order):
        noisy = None if self.noisy_point_clouds is None else self.noisy_point_clouds[order]
        return self.point_clouds[order], self.labels[order], noisy

    def shuffle_data(self, seed=None):
        """Re-orders the examples (clouds, labels and noisy copies alike)."""
        self.point_clouds, self.labels, self.noisy_point_clouds = self._rows(_legacy_permutation(self.num_examples, seed))
        return self

    def shuffle_points(self, seed=None):
        """Re-orders the points inside every cloud.  The reference keeps ONE index vector and shuffles it again for each
        example, so example i's order is the composition of i + 1 shuffles: the orders are drawn first, then applied at once."""
        if seed is not None:
            np.random.seed(seed)
        walk = np.arange(self.n_points)
        order = np.empty((self.num_examples, self.n_points), dtype=np.intp)
        for row in order:
            np.random.shuffle(walk)
            row[:] = walk
        self.point_clouds = np.take_along_axis(self.point_clouds, order[:, :, None], axis=1)
        if self.noisy_point_clouds is not None:
            self.noisy_point_clouds = np.take_along_axis(self.noisy_point_clouds, order[:, :, None], axis=1)
        return self

    def next_batch(self, batch_size, seed=None):
        """(point_clouds, labels, noisy_point_clouds | None) of the next batch_size examples; an epoch that cannot supply a
        full batch ends: the data are re-shuffled and the batch is taken from the start."""
        if self._cursor + batch_size > self.num_examples:
            self.epochs_completed += 1
            self.shuffle_data(seed)
            self._cursor = 0
        window = slice(self._cursor, self._cursor + batch_size)
        self._cursor += batch_size
        return self._rows(window)

    def full_epoch_data(self, shuffle=True, seed=None):
        """The whole set at once (optionally in a fresh random order; the set's own order is untouched)."""
        order = _legacy_permutation(self.num_examples, seed) if shuffle else np.arange(self.num_examples)
        return self._rows(order)

    def merge(self, other_data_set):
        """Appends another set's examples; the epoch state starts over."""
        self.epochs_completed, self._cursor = 0, 0
        self.point_clouds = np.concatenate([self.point_clouds, other_data_set.point_clouds], axis=0)
        self.labels = np.concatenate([self.labels.reshape(self.num_examples), other_data_set.labels.reshape(other_data_set.num_examples)])
        if self.noisy_point_clouds is not None:
            self.noisy_point_clouds = np.concatenate([self.noisy_point_clouds, other_data_set.noisy_point_clouds], axis=0)
        self.num_examples = self.point_clouds.shape[0]
        return self

    # (name the reference's scripts may poke at)
    @property
    def _index_in_epoch(self):
        return self._cursor


def _folder_clouds(top_dir, n_threads, file_ending, verbose):
    names = list(files_in_subdirs(top_dir, file_ending))
    return load_point_clouds_from_filenames(names, n_threads, loader=pc_loader, verbose=verbose)


def load_all_point_clouds_under_folder(top_dir, n_threads=20, file_ending=".ply", verbose=False):
    """One PointCloudDataSet over every cloud below top_dir, labelled "<syn_id>_<model_id>"  (in_out.py:176-182)."""
    pclouds, model_ids, syn_ids = _folder_clouds(top_dir, n_threads, file_ending, verbose)
    return PointCloudDataSet(pclouds, labels=syn_ids + "_" + model_ids, init_shuffle=False)


def load_and_split_all_point_clouds_under_folder(top_dir, n_threads=20, file_ending=".ply", split=(0.85, 0.05, 0.10), seed=42,
                                                 verbose=False):
    """(train, val, test) PointCloudDataSets: one seeded permutation applied to clouds and labels  (in_out.py:185-217)."""
    pclouds, model_ids, syn_ids = _folder_clouds(top_dir, n_threads, file_ending, verbose)
    labels = syn_ids + "_" + model_ids
    *clouds, perm = split_data(pclouds, split, seed)
    *names, _ = split_data(labels, split, seed, perm)
    return tuple(PointCloudDataSet(pc, labels=lb, init_shuffle=False) for pc, lb in zip(clouds, names))


# ----------------------------------------------------------------------------------------------------- ModelNet40 (HDF5)
def _h5_shard(path):
    """(data (clouds, 2048, 3) float32, label (clouds, 1)) of one `ply_data_*.h5` shard -- through h5py like the reference
    (registration/data/modelnet_loader_torch.py:26-30).  h5py is not part of this image: pass `shard_reader=` to ModelNetCls
    to read shards converted to another container (tests use .npz twins of the same two arrays)."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("ModelNetCls reads HDF5 shards through h5py (not installed); convert the shards and pass "
                          "shard_reader=, e.g. lambda p: (np.load(p)['data'], np.load(p)['label'])") from e
    with h5py.File(path, "r") as f:
        return f["data"][:], f["label"][:]


class ModelNetCls(object):
    """ModelNet40 clouds from the `modelnet40_ply_hdf5_2048` shards as a map-style dataset (__getitem__ / __len__ for
    torch.utils.data.DataLoader): constructor arguments, attributes (points, labels, num_points, classes, class_to_idx, shapes)
    and item convention of registration/data/modelnet_loader_torch.py:32-127.  Every item is a fresh random subset / order of
    its cloud's first num_points points (numpy's global generator, as the reference).  The shards must be on disk under
    base_dir/folder (no network here: download=True only changes the error text).
    shard_reader (new): callable path -> (data, label); default = h5py."""

    def __init__(self, num_points, transforms, train, download=False, cinfo=None, folder="modelnet10_hdf5_2048", url=None,
                 include_shapes=False, base_dir=None, shard_reader=None):
        base_dir = base_dir or os.getcwd()
        self.transforms, self.folder, self.train = transforms, folder, train
        self.data_dir = osp.join(base_dir, folder)
        if not osp.isdir(self.data_dir):
            hint = " (downloading is not supported: fetch %s there)" % url if download else ""
            raise FileNotFoundError("ModelNet shards not found under %s%s" % (self.data_dir, hint))
        split = "train" if train else "test"
        with open(osp.join(self.data_dir, split + "_files.txt")) as f:
            # the list files name the shards relative to a "data/" folder: "data/modelnet40_ply_hdf5_2048/ply_data_train0.h5"
            self.files = [line.rstrip()[len("data/"):] for line in f if line.strip()]
        read = shard_reader or _h5_shard
        shards = [read(osp.join(base_dir, name)) for name in self.files]
        self.points = np.concatenate([d for d, _ in shards], axis=0)
        labels = np.concatenate([np.asarray(l) for _, l in shards], axis=0)
        self.labels = labels[:, None] if labels.ndim == 1 else labels
        self.set_num_points(num_points)
        self.classes, self.class_to_idx = cinfo if cinfo is not None else (None, None)
        self.include_shapes = include_shapes
        self.shapes = []
        if include_shapes:
            for n in range(len(self.files)):
                with open(osp.join(self.data_dir, "ply_data_%s_%d_id2file.json" % (split, n))) as f:
                    self.shapes.extend(json.load(f))

    def __len__(self):
        return self.points.shape[0]

    def __getitem__(self, idx):
        import torch

        order = np.random.permutation(self.num_points)  # (= arange + shuffle on the global generator)
        cloud = self.points[idx, order].copy()
        if self.transforms is not None:
            cloud = self.transforms(cloud)
        label = torch.from_numpy(self.labels[idx]).type(torch.LongTensor)
        return (cloud, label, self.shapes[idx]) if self.include_shapes else (cloud, label)

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
            ring.release(i)      # the next load(i) must wait for this replay (an unreleased slot is overwritten under it)
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
