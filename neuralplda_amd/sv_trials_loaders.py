"""Batch interface of the hot path — counterpart of utils/sv_trials_loaders.py:371-437.

Same function names, arguments and outputs as the reference; what changes is where the work happens:

* `combine_trials_and_get_loader` / `get_trials_loaders_dict` consume the torch / numpy RNG streams
  exactly like the reference (np.random.rand per file, torch DataLoader shuffle), so the kept set and
  the batch order are identical for identical seeds — but a batch is collated with ONE vectorised
  index instead of B `TensorDataset.__getitem__` calls.
* `load_xvec_trials_from_numbatch` / `_from_idbatch` keep the x-vectors in a device-resident
  (N_utt, 512) matrix built once per `mega_dict` and replace the per-pair Python dict look-ups
  (utils/sv_trials_loaders.py:420-423, the reference's true bottleneck at 1.8e4 pairs/s) by the HIP
  gather kernel nplda_gather_rows_f32.  When the caller asks for CPU tensors (`device` is the CPU, as
  utils/scorefile_generator.py:34 does) the rows are index-selected on the host — that is data
  movement for a host consumer, not a compute fallback.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

__all__ = ["combine_trials_and_get_loader", "get_trials_loaders_dict", "load_xvec_trials_from_numbatch",
           "load_xvec_trials_from_idbatch", "XvectorTable", "xvector_table", "TrialIndexDataset"]


# ---------------------------------------------------------------------------------------------------
# trial lists -> loaders
# ---------------------------------------------------------------------------------------------------

class TrialIndexDataset(Dataset):
    """len() == number of kept trials; item i is just i — batches are materialised by the collate_fn."""

    def __init__(self, x1, x2, l):
        self.x1, self.x2, self.l = x1, x2, l
        self.dropped = 0

    def __len__(self):
        return self.x1.shape[0]

    def __getitem__(self, i):
        return i

    def collate(self, idxs):
        ii = torch.as_tensor(idxs, dtype=torch.int64)
        return self.x1[ii], self.x2[ii], self.l[ii]


_IDBLOBS = {}


def _dict_blob(id_to_num_dict):
    """textio.IdBlob of an {utt_id: num} dict, cached per dict object (rebuilt when its size changes)."""
    from . import textio
    k = id(id_to_num_dict)
    hit = _IDBLOBS.get(k)
    if hit is None or hit[0] != len(id_to_num_dict):
        if len(_IDBLOBS) > 8:
            _IDBLOBS.clear()
        hit = (len(id_to_num_dict), textio.IdBlob.from_dict(id_to_num_dict))
        _IDBLOBS[k] = hit
    return hit[1]


def _read_trials(f, id_to_num_dict, strip_ext_col2):
    """np.genfromtxt + the per-trial id mapping loop of utils/sv_trials_loaders.py:377-383 (:400-406) as one native
    pass over the file (nplda_text_lookup); rows with unknown ids / bad labels are dropped silently by the
    reference — here they are counted (returned)."""
    from . import textio
    with open(f, "rb") as fh:
        text = fh.read()
    rows, _ = textio.scan(text)
    x1, x2, l, _, _ = textio.lookup(text, _dict_blob(id_to_num_dict), 0, textio.RAW,
                                    textio.SPLITEXT if strip_ext_col2 else textio.RAW, label_col=2, rows=rows)
    return (torch.from_numpy(x1.copy()), torch.from_numpy(x2.copy()), torch.from_numpy(l.copy()), rows - len(x1))


def _tensor_version(t):
    """A tensor's in-place version counter, 0 for an inference tensor (which has none: reading `_version` raises, and it
    cannot be modified in place outside inference mode either) — the guard ops.cohort_prepare uses."""
    return 0 if t.is_inference() else t._version


def _seed_broadcaster(device, group=None):
    """seed -> rank 0's seed (one 8-byte broadcast on `group`); None when there is nothing to synchronise with."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return None

    def sync(seed):
        on_dev = dist.get_backend(group) == "nccl"
        t = torch.tensor([seed], dtype=torch.int64, device=device if on_dev else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return int(t.item())
    return sync


class TrialLoader(DataLoader):
    """DataLoader(dataset, batch_size, shuffle=True) over a TrialIndexDataset with a vectorised iterator: the batches are
    slices of ONE permutation instead of 2048 Python ints per batch going through sampler -> list -> collate (0.4 ms per
    batch, four times the fused training step).  The permutation is drawn exactly as torch's own machinery would draw it
    — the iterator's base-seed draw, then RandomSampler's seed draw, then randperm on a fresh generator — so the batch
    sequence under a given torch.manual_seed is the one the reference's DataLoader yields (pinned by the G8 fixture and
    tests/test_host_logic.py)."""

    def __iter__(self):
        ds = self.dataset
        if not isinstance(ds, TrialIndexDataset) or self.num_workers != 0 or self.drop_last:
            return super().__iter__()
        return self._fast_iter(ds)

    def _fast_iter(self, ds):
        n = len(ds)
        torch.empty((), dtype=torch.int64).random_()            # _BaseDataLoaderIter's base seed (consumed, unused)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())  # RandomSampler.__iter__
        gen = torch.Generator()
        gen.manual_seed(seed)
        perm = torch.randperm(n, generator=gen)
        bs = self.batch_size
        for lo in range(0, n, bs):
            ii = perm[lo:lo + bs]
            yield ds.x1[ii], ds.x2[ii], ds.l[ii]

    def _device_epoch_arrays(self, device, num_to_row, permute=True, seed_sync=None):
        """One epoch's permuted index / label arrays on `device` (the draws of __iter__: same permutation).
        permute=False: the same two draws from the global generator (its state moves on exactly as an iteration would move
        it) but the trials in file order — for a consumer to whom the order means nothing (validate(): every metric is a
        function of the SET of (score, label) pairs).  The host-side randperm of 1 M indices is 8 - 60 ms, the rest of a
        1 M-trial validation pass 6 ms."""
        ds = self.dataset
        if not isinstance(ds, TrialIndexDataset) or self.num_workers != 0 or self.drop_last:
            raise TypeError("device_batches needs the vectorised TrialIndexDataset path")
        n = len(ds)
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        if seed_sync is not None:  # data parallel: every rank walks rank 0's epoch whatever its own generator's state
            seed = int(seed_sync(seed))
        gen = torch.Generator()
        gen.manual_seed(seed)
        device = torch.device(device)
        if not permute:
            if device.type == "cuda":  # the columns stay on the device across calls: copied on a cache miss only
                key = (str(device), ds.x1.data_ptr(), ds.x2.data_ptr(), ds.l.data_ptr(), n)
                cache = getattr(self, "_dev_columns", None)
                if cache is None or cache[0] != key:
                    cache = self._dev_columns = (key, ds.x1.to(device), ds.x2.to(device), ds.l.to(device))
                e1, e2, el = cache[1], cache[2], cache[3]
            else:
                e1, e2, el = ds.x1.to(device), ds.x2.to(device), ds.l.to(device)
            if num_to_row is not None:
                # (the mapped columns of a resident list are kept: the mapping is two index launches and two
                # synchronising min() read-backs per call, and validate() walks the same list every epoch)
                # (an inference tensor has no version counter — it cannot be written in place outside inference mode
                # either; the cache holds the map itself and compares with `is`, so a freed map's address reused by
                # another tensor can never hit)
                mkey = (key if device.type == "cuda" else None, _tensor_version(num_to_row), num_to_row.numel())
                mc = getattr(self, "_dev_columns_mapped", None)
                if mc is not None and mkey[0] is not None and mc[0] == mkey and mc[3] is num_to_row:
                    return n, mc[1], mc[2], el
                e1, e2 = num_to_row[e1.long()], num_to_row[e2.long()]
                if n and (int(e1.min()) < 0 or int(e2.min()) < 0):
                    raise KeyError("trial index refers to an utterance that is not in mega_dict")
                if mkey[0] is not None:
                    self._dev_columns_mapped = (mkey, e1, e2, num_to_row)
            return n, e1, e2, el
        perm = torch.randperm(n, generator=gen)
        if device.type == "cuda":
            # the dataset's three columns live on the device across epochs; an epoch sends its permutation and gathers there
            # (the host-side gathers were two thirds of an epoch's set-up time)
            key = (str(device), ds.x1.data_ptr(), ds.x2.data_ptr(), ds.l.data_ptr(), n)
            cache = getattr(self, "_dev_columns", None)
            if cache is None or cache[0] != key:
                cache = self._dev_columns = (key, ds.x1.to(device), ds.x2.to(device), ds.l.to(device))
            pd = perm.to(device, non_blocking=True)
            e1, e2, el = cache[1][pd], cache[2][pd], cache[3][pd]
        else:
            e1 = ds.x1[perm].to(device, non_blocking=True)
            e2 = ds.x2[perm].to(device, non_blocking=True)
            el = ds.l[perm].to(device, non_blocking=True)
        if num_to_row is not None:
            e1, e2 = num_to_row[e1.long()], num_to_row[e2.long()]
            if n and (int(e1.min()) < 0 or int(e2.min()) < 0):
                raise KeyError("trial index refers to an utterance that is not in mega_dict")
        return n, e1, e2, el

    def _pack_records(self, n, e1, e2, el, device):
        """One contiguous record per full batch — [rows1 (int64) | rows2 (int64) | labels (float32)], back to back."""
        bs = self.batch_size
        if bs % 2:
            # record k starts at 20 * bs * k bytes: with an odd batch size every other record's int64 fields are misaligned
            raise ValueError("packed batch records need an even batch_size (use device_batches(pack=False))")
        nb = n // bs
        rec = torch.empty((nb, 20 * bs), dtype=torch.uint8, device=device)
        if nb:
            rec[:, :8 * bs].view(torch.int64).copy_(e1[:nb * bs].view(nb, bs))
            rec[:, 8 * bs:16 * bs].view(torch.int64).copy_(e2[:nb * bs].view(nb, bs))
            rec[:, 16 * bs:].view(torch.float32).copy_(el[:nb * bs].float().view(nb, bs))
        return rec

    def device_epoch(self, device, num_to_row=None):
        """The epoch as (records, tail): `records` a (full batches, 20 * batch_size) uint8 device tensor of packed batch
        records, `tail` the last partial batch as (rows1, rows2, labels) device views or None.  A consumer that walks the
        records on the device (FusedTrainStep.begin_epoch / step_record) needs no per-batch copy at all."""
        n, e1, e2, el = self._device_epoch_arrays(device, num_to_row)
        bs = self.batch_size
        lo = n // bs * bs
        tail = (e1[lo:], e2[lo:], el[lo:]) if lo < n else None
        return self._pack_records(n, e1, e2, el, device), tail

    def device_columns(self, device, num_to_row=None):
        """The whole trial list in FILE order as three device arrays (n, rows1, rows2, labels) — for a consumer to whom
        neither order nor batching means anything (validate(): its metrics are functions of the set of (score, label)
        pairs, so it scores in chunks sized for the kernels, not for the loader).  The global RNG moves on exactly as one
        iteration of the loader would move it."""
        return self._device_epoch_arrays(device, num_to_row, permute=False)

    def device_columns_distinct(self, device, num_to_row=None):
        """device_columns() plus the trial list's DISTINCT table rows: (n, rows1, rows2, labels, urows, j1, j2) with
        `urows` the sorted distinct values of rows1 and rows2 together and rows1 == urows[j1], rows2 == urows[j2].  A trial
        list names each utterance many times (a 1 M-trial validation list over 200 k utterances: ten times), so a consumer
        may embed `urows` once and score by (j1, j2) — validate()'s embed-once pass.  The distinct set is a function of the
        (fixed) trial list and the map: it is formed once (one device sort) and kept with the resident columns."""
        n, e1, e2, el = self.device_columns(device, num_to_row)
        resident = torch.device(device).type == "cuda" and getattr(self, "_dev_columns", None) is not None
        key = (self._dev_columns[0] if resident else None, None if num_to_row is None else
               (_tensor_version(num_to_row), num_to_row.numel()))
        cache = getattr(self, "_dev_distinct", None)
        if cache is None or cache[0] != key or key[0] is None or cache[4] is not num_to_row:
            urows, inv = torch.unique(torch.cat([e1.long(), e2.long()]), return_inverse=True)
            cache = self._dev_distinct = (key, urows, inv[:n].contiguous(), inv[n:].contiguous(), num_to_row)
        return n, e1, e2, el, cache[1], cache[2], cache[3]

    def device_batches(self, device, num_to_row=None, pack=False, permute=True, shard=None, group=None):
        """The same epoch (same permutation, same RNG draws) with the three index arrays moved to `device` ONCE and the
        batches yielded as device views: three host-to-device copies per epoch instead of three per batch.
        `num_to_row`: optional int64 device map applied to both index columns (trial number -> x-vector table row);
        a negative entry (unknown utterance) raises KeyError like load_xvec_trials_from_numbatch.
        pack=True also yields, per full batch, the batch as one contiguous uint8 record (None for a last partial batch).
        permute=False: file order (see _device_epoch_arrays) for order-free consumers.
        shard=(rank, world) (data parallel; every rank must draw the same epoch, i.e. share the RNG state): yields this
        rank's contiguous slice of every GLOBAL batch and, fourth, the global batch's [N_t, N_n] as a device float64
        tensor — all the one-collective training step needs to know about the other ranks' shards (the counts of the
        whole epoch are formed in one pass).  The loader must be the GLOBAL, unsharded one (the slicing happens here);
        when torch.distributed is initialised the epoch's permutation seed is broadcast from rank 0 of `group`, so a rank
        whose generator has drifted (a rank-0-only validate(), say) still walks the same epoch as the others."""
        n, e1, e2, el = self._device_epoch_arrays(device, num_to_row, permute,
                                                  seed_sync=_seed_broadcaster(device, group) if shard is not None else None)
        bs = self.batch_size
        if shard is not None:
            if pack:
                raise ValueError("device_batches: packed records describe whole batches; shard=(rank, world) yields views")
            rank, world = shard
            nb = (n + bs - 1) // bs
            csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=el.device), el.double().cumsum(0)])
            edges = torch.clamp(torch.arange(nb + 1, device=el.device) * bs, max=n)
            nt = csum[edges[1:]] - csum[edges[:-1]]
            counts = torch.stack([nt, (edges[1:] - edges[:-1]).double() - nt], 1).contiguous()
            for k, lo in enumerate(range(0, n, bs)):
                B = min(bs, n - lo)
                chunk = (B + world - 1) // world  # dist.shard_bounds: chunks of ceil(B / world), the last short or empty
                a = lo + min(rank * chunk, B)
                b = lo + min(rank * chunk + chunk, B)
                yield e1[a:b], e2[a:b], el[a:b], counts[k]
            return
        if pack and bs % 2:
            pack = None  # no aligned record layout for an odd batch size: the consumer gets record=None and copies the views
        if pack is None:
            for lo in range(0, n, bs):
                yield e1[lo:lo + bs], e2[lo:lo + bs], el[lo:lo + bs], None
            return
        if pack:
            # a consumer with static input buffers (FusedTrainStep.step_rows) stages a batch with ONE device copy instead of three
            rec = self._pack_records(n, e1, e2, el, device)
            nb = n // bs
            for k, lo in enumerate(range(0, n, bs)):
                yield e1[lo:lo + bs], e2[lo:lo + bs], el[lo:lo + bs], (rec[k] if k < nb else None)
            return
        for lo in range(0, n, bs):
            yield e1[lo:lo + bs], e2[lo:lo + bs], el[lo:lo + bs]


def _loader(ds, batch_size):
    return TrialLoader(ds, batch_size=batch_size, shuffle=True, collate_fn=ds.collate)


def combine_trials_and_get_loader(trials_key_files_list, id_to_num_dict, subsample_factors=None, batch_size=2048,
                                  subset=0):
    """utils/sv_trials_loaders.py:371-392."""
    if subsample_factors is None:
        subsample_factors = [1 for w in trials_key_files_list]
    parts, dropped = [], 0
    for f, sf in zip(trials_key_files_list, subsample_factors):
        x1, x2, l, d = _read_trials(f, id_to_num_dict, strip_ext_col2=False)
        dropped += d
        inds = torch.from_numpy(np.arange(len(x1))[np.random.rand(len(x1)) < sf])
        parts.append((x1[inds], x2[inds], l[inds]))
    x1 = torch.cat([p[0] for p in parts]) if parts else torch.zeros(0, dtype=torch.int64)
    x2 = torch.cat([p[1] for p in parts]) if parts else torch.zeros(0, dtype=torch.int64)
    l = torch.cat([p[2] for p in parts]) if parts else torch.zeros(0)
    if subset > 0:
        inds = torch.from_numpy(np.arange(len(x1))[np.random.rand(len(x1)) < subset])
        x1, x2, l = x1[inds], x2[inds], l[inds]
    ds = TrialIndexDataset(x1, x2, l)
    ds.dropped = dropped
    return _loader(ds, batch_size)


def get_trials_loaders_dict(trials_key_files_list, id_to_num_dict, subsample_factors=None, batch_size=2048, subset=0):
    """utils/sv_trials_loaders.py:394-415 (column 2 loses its file extension, key = file basename sans ext)."""
    trials_loaders_dict = {}
    if subsample_factors is None:
        subsample_factors = [1 for w in trials_key_files_list]
    for f, sf in zip(trials_key_files_list, subsample_factors):
        x1, x2, l, d = _read_trials(f, id_to_num_dict, strip_ext_col2=True)
        inds = torch.from_numpy(np.arange(len(x1))[np.random.rand(len(x1)) < sf])
        x1, x2, l = x1[inds], x2[inds], l[inds]
        if subset > 0:
            inds = torch.from_numpy(np.arange(len(x1))[np.random.rand(len(x1)) < subset])
            x1, x2, l = x1[inds], x2[inds], l[inds]
        ds = TrialIndexDataset(x1, x2, l)
        ds.dropped = d
        trials_loaders_dict[os.path.splitext(os.path.basename(f))[0]] = _loader(ds, batch_size)
    return trials_loaders_dict


# ---------------------------------------------------------------------------------------------------
# x-vector table (the "mega dict" {utt_id: float32[512]}, dataprep_sre.py:152-167) resident on the device
# ---------------------------------------------------------------------------------------------------

class XvectorTable:
    """Row-major (N_utt, D) float32 matrix of x-vectors with id -> row maps, host and device copies.

    Built from a mega dict {utt_id: float32[D]} (the pickle dataprep_sre.py:152-167 writes) or — without ever
    materialising that dict — straight from the Kaldi archives it was made from (`from_ark`, `from_scp`: the vectors go
    archive -> one pinned host matrix -> device).  The table is accepted wherever the loaders / score generators /
    train() take `mega_dict`: it iterates, indexes and reports its length like the dict of the reference."""

    def __init__(self, mega_dict):
        self.ids = list(mega_dict.keys())
        self.row_of = {u: i for i, u in enumerate(self.ids)}
        self.host = np.ascontiguousarray(np.stack([np.asarray(mega_dict[u], dtype=np.float32) for u in self.ids])
                                         if self.ids else np.zeros((0, 0), np.float32))
        self._dev = {}
        self._num_maps = {}
        self._idblob = None
        self._pinned = None

    @classmethod
    def from_matrix(cls, ids, mat, pinned=None):
        """ids (N) + an (N, D) float32 matrix (numpy; `pinned`: the CPU torch tensor that owns its memory)."""
        self = cls.__new__(cls)
        self.ids = list(ids)
        if mat.shape[0] != len(self.ids):
            raise ValueError("one id per row")
        self.row_of = {u: i for i, u in enumerate(self.ids)}
        if len(self.row_of) != len(self.ids):  # a repeated key: the last occurrence wins, as dict.update does
            keep = sorted(self.row_of.values())
            self.ids = [self.ids[i] for i in keep]
            mat = np.ascontiguousarray(mat[keep])
            self.row_of = {u: i for i, u in enumerate(self.ids)}
            pinned = None
        self.host = mat
        self._dev, self._num_maps, self._idblob, self._pinned = {}, {}, None, pinned
        return self

    @staticmethod
    def _pinned_buffer(n, dim):
        t = torch.empty((n, dim), dtype=torch.float32)
        if torch.cuda.is_available():
            try:
                t = t.pin_memory()
            except RuntimeError:
                pass
        return t

    @classmethod
    def from_ark(cls, *ark_paths):
        """Binary Kaldi vector archives -> table (rows in archive order; dataprep_sre.py:152-167 without kaldi_io, the
        per-utterance arrays, the dict and the pickle)."""
        from . import kaldi_format
        parts = [kaldi_format.load_vector_ark(p) for p in ark_paths]
        return cls._from_parts(parts)

    @classmethod
    def from_scp(cls, *scp_paths):
        """Kaldi scp files ('utt ark:offset' lines) -> table, every referenced archive memory-mapped once."""
        from . import kaldi_format
        parts = [kaldi_format.load_vector_scp(p) for p in scp_paths]
        return cls._from_parts(parts)

    @classmethod
    def _from_parts(cls, parts):
        parts = [(k, m) for k, m in parts if len(k)]
        if not parts:
            return cls.from_matrix([], np.zeros((0, 0), np.float32))
        n, dim = sum(len(k) for k, _ in parts), parts[0][1].shape[1]
        buf = cls._pinned_buffer(n, dim)
        host = buf.numpy()
        ids, lo = [], 0
        for k, m in parts:
            host[lo:lo + len(k)] = m
            ids += k
            lo += len(k)
        return cls.from_matrix(ids, host, pinned=buf)

    # -- the dict protocol the reference's scripts use on mega_xvec_dict (iteration order = row order)
    def __len__(self):
        return len(self.ids)

    def __iter__(self):
        return iter(self.ids)

    def keys(self):
        return self.ids

    def __contains__(self, u):
        return u in self.row_of

    def __getitem__(self, u):
        return self.host[self.row_of[u]]

    @property
    def idblob(self):
        """The id -> row table in the layout the native text routines read (built once)."""
        if self._idblob is None:
            from . import textio
            self._idblob = textio.IdBlob(self.ids)
        return self._idblob

    @property
    def dim(self):
        return self.host.shape[1]

    def on(self, device):
        device = torch.device(device)
        key = (device.type, device.index if device.index is not None else (torch.cuda.current_device()
                                                                           if device.type == "cuda" else -1))
        if key not in self._dev:
            src = self._pinned if self._pinned is not None else torch.from_numpy(self.host)
            self._dev[key] = src.to(device, non_blocking=self._pinned is not None)
        return self._dev[key]

    def rows_from_nums(self, num_to_id_dict):
        """int64 numpy map num -> table row for a {num: utt_id} dict (cached per dict object)."""
        k = id(num_to_id_dict)
        hit = self._num_maps.get(k)
        if hit is None or hit[0] != len(num_to_id_dict):
            n = (max(num_to_id_dict.keys()) + 1) if len(num_to_id_dict) else 0
            m = np.full(n, -1, dtype=np.int64)
            for num, u in num_to_id_dict.items():
                r = self.row_of.get(u)
                if r is not None:
                    m[int(num)] = r
            hit = (len(num_to_id_dict), m, {})
            self._num_maps[k] = hit
        return hit

    def gather(self, rows, device):
        """rows: int64 numpy array or torch tensor of table rows -> (B, D) float32 on `device`."""
        device = torch.device(device)
        if device.type == "cuda":
            from . import ops
            table = self.on(device)
            if not isinstance(rows, torch.Tensor):
                rows = torch.from_numpy(np.ascontiguousarray(rows))
            return ops.gather_rows(table, rows.to(device, non_blocking=True))
        rows = rows.cpu().numpy() if isinstance(rows, torch.Tensor) else np.asarray(rows)
        if rows.size and (rows.min() < 0 or rows.max() >= self.host.shape[0]):
            raise KeyError("trial refers to an utterance that is not in the x-vector table")
        return torch.from_numpy(self.host[rows]).float().to(device)


_TABLES = {}


def xvector_table(mega_dict):
    """The (cached) XvectorTable of a mega dict; rebuilt if the dict object or its size changed.  An XvectorTable
    (e.g. XvectorTable.from_scp) passes through."""
    if isinstance(mega_dict, XvectorTable):
        return mega_dict
    k = id(mega_dict)
    hit = _TABLES.get(k)
    if hit is None or hit[0] != len(mega_dict):
        if len(_TABLES) > 8:
            _TABLES.clear()
        hit = (len(mega_dict), XvectorTable(mega_dict))
        _TABLES[k] = hit
    return hit[1]


def load_xvec_trials_from_numbatch(mega_dict, num_to_id_dict, data1, data2, device):
    """utils/sv_trials_loaders.py:418-426: int index batches -> 2 x (B, 512) float32 on `device`."""
    tab = xvector_table(mega_dict)
    _, m, devmaps = tab.rows_from_nums(num_to_id_dict)
    device = torch.device(device)
    if device.type == "cuda" and isinstance(data1, torch.Tensor) and data1.is_cuda:
        key = (data1.device.type, data1.device.index)
        if key not in devmaps:
            devmaps[key] = torch.from_numpy(m).to(data1.device)
        # one launch maps both index columns to table rows and gathers both sides; the reference's dict look-up raises
        # KeyError for an unknown number, here a flag word the kernel raises is read back (ops.gather_pairs_mapped)
        from . import ops
        if device == data1.device or device.index is None:
            return ops.gather_pairs_mapped(tab.on(data1.device), devmaps[key], data1, data2)
        x1, x2 = ops.gather_pairs_mapped(tab.on(data1.device), devmaps[key], data1, data2)
        return x1.to(device), x2.to(device)
    d1 = data1.cpu().numpy() if isinstance(data1, torch.Tensor) else np.asarray(data1)
    d2 = data2.cpu().numpy() if isinstance(data2, torch.Tensor) else np.asarray(data2)
    r1, r2 = m[d1.reshape(-1).astype(np.int64)], m[d2.reshape(-1).astype(np.int64)]
    if (r1.size and r1.min() < 0) or (r2.size and r2.min() < 0):
        raise KeyError("trial index refers to an utterance that is not in mega_dict")
    return tab.gather(r1, device), tab.gather(r2, device)


def _ids_to_rows(tab, ids):
    try:
        return np.fromiter((tab.row_of[os.path.splitext(os.path.basename(d))[0]] for d in ids), dtype=np.int64,
                           count=len(ids))
    except KeyError as e:
        raise KeyError(f"utterance {e.args[0]!r} is not in mega_dict") from None


def load_xvec_trials_from_idbatch(mega_dict, trials, device):
    """utils/sv_trials_loaders.py:429-437: (B, >=2) array of id strings (directory and extension are
    stripped from both columns) -> 2 x (B, 512) float32 on `device`.  An empty batch gives (0, 512)."""
    tab = xvector_table(mega_dict)
    trials = np.asarray(trials)
    if trials.size == 0:
        e = torch.zeros((0, tab.dim), dtype=torch.float32, device=device)
        return e, e.clone()
    trials = trials.reshape(-1, trials.shape[-1])
    return (tab.gather(_ids_to_rows(tab, trials[:, 0]), device), tab.gather(_ids_to_rows(tab, trials[:, 1]), device))
