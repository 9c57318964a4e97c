"""Host-side input pipeline with the surface of `openrec.tf2.data.Dataset`
(openrec/tf2/data/dataset.py:87-176, utils.py:6-214): an in-memory interaction
index plus pairwise / pointwise / evaluation batch iterators.

It feeds the fused device step, so it is built differently from the reference
(which assembles Python lists sample by sample inside spawned processes and
pickles them through a queue): the index is a NumPy CSR over (user, item),
batches are produced as int32 arrays in-process or by background threads.

Exact-sequence parity: driven in-process with a seed, `pairwise`,
`stratified_pointwise` and `per_pos_stratified_pointwise` consume the CPython
`random` stream in the same order as the reference's generators
 and therefore yield identical samples -- pinned against
the reference's own code in tests/test_datalayer.py.  With
`num_parallel_calls > 1` each producer has its own OS-seeded stream, like the
reference's spawned processes (whose output is not reproducible either).
"""
from __future__ import annotations

import queue
import random
import threading

import numpy as np


class InteractionIndex:
    """CSR index of the positive (and, if labelled, negative) interactions."""

    def __init__(self, raw_data, total_users, total_items, implicit_negative=True, num_negatives=None,
                 sortby=None, asc=True):
        if not isinstance(raw_data, np.ndarray):
            raise TypeError("Unsupported data input schema. Please use structured numpy array.")
        self.raw = raw_data
        self.total_users, self.total_items = int(total_users), int(total_items)
        self.implicit_negative, self.num_negatives = implicit_negative, num_negatives
        users = np.asarray(raw_data["user_id"], np.int64)
        items = np.asarray(raw_data["item_id"], np.int64)
        if implicit_negative:
            pos = np.ones(len(raw_data), bool)
        else:
            pos = np.asarray(raw_data["label"]) > 0
        self._pos_ptr, self._pos_items, self._pos_rec = self._csr(users[pos], items[pos], np.nonzero(pos)[0])
        # membership: sorted keys user * total_items + item
        self._pos_keys = np.unique(users[pos] * self.total_items + items[pos])
        self._pos_set = set(self._pos_keys.tolist())
        self._neg_ptr = self._neg_items = None
        if not implicit_negative:
            self._neg_ptr, self._neg_items, _ = self._csr(users[~pos], items[~pos], np.nonzero(~pos)[0])
        elif num_negatives is not None:
            # pre-sampled negatives per user with positives (NumPy global RNG, like the reference)
            ptr = [0]; out = []
            for u in range(self.total_users):
                if self._pos_ptr[u + 1] > self._pos_ptr[u]:
                    perm = np.random.permutation(self.total_items)
                    mine = self.positive_items(u)
                    cand = perm[~np.isin(perm, mine)][:num_negatives]
                    out.append(cand)
                    ptr.append(ptr[-1] + len(cand))
                else:
                    ptr.append(ptr[-1])
            self._neg_ptr = np.asarray(ptr, np.int64)
            self._neg_items = np.concatenate(out) if out else np.zeros(0, np.int64)
        self.sortby, self.asc = sortby, asc
        # first-seen order of users (the reference iterates a dict keyed in insertion order)
        _, first = np.unique(users[pos], return_index=True)
        self._user_order = users[pos][np.sort(first)]

    def _csr(self, users, items, rec):
        order = np.argsort(users, kind="stable")
        u, i, r = users[order], items[order], rec[order]
        # the reference keeps the LAST record index of a repeated (user, item) pair
        key = u * self.total_items + i
        _, last = np.unique(key[::-1], return_index=True)
        keep = np.sort(len(key) - 1 - last)
        u, i, r = u[keep], i[keep], r[keep]
        ptr = np.zeros(self.total_users + 1, np.int64)
        np.add.at(ptr, u + 1, 1)
        return np.cumsum(ptr), i, r

    # ---- queries ---------------------------------------------------------
    def is_positive(self, user, item):
        return int(user) * self.total_items + int(item) in self._pos_set

    def has_positives(self, user):
        return self._pos_ptr[user + 1] > self._pos_ptr[user]

    def positive_items(self, user, sort=False):
        lo, hi = self._pos_ptr[user], self._pos_ptr[user + 1]
        items = self._pos_items[lo:hi]
        if sort:
            assert self.sortby is not None, "sortby key is not specified."
            keys = self.raw[self.sortby][self._pos_rec[lo:hi]]
            order = np.argsort(keys, kind="stable")
            items = items[order] if self.asc else items[order[::-1]]
        return items

    def negative_items(self, user):
        if self._neg_ptr is not None:
            return self._neg_items[self._neg_ptr[user]:self._neg_ptr[user + 1]]
        mask = np.ones(self.total_items, bool)
        mask[self.positive_items(user)] = False
        return np.nonzero(mask)[0]

    def contain_negatives(self):
        return not (self.implicit_negative and self.num_negatives is None)

    def warm_users(self, threshold=1):
        cnt = np.diff(self._pos_ptr)
        return [int(u) for u in self._user_order if cnt[u] >= threshold]

    def total_records(self):
        return len(self.raw)


class _Sampler:
    """One RNG stream over an index: the draw order of the reference's generators."""

    def __init__(self, index, rng):
        self.ix, self.rng = index, rng
        self._rand_ids = []

    def next_random_record(self):
        if not self._rand_ids:                      # one epoch = one permutation, popped from the end
            self._rand_ids = list(range(self.ix.total_records()))
            self.rng.shuffle(self._rand_ids)
        return self.ix.raw[self._rand_ids.pop()]

    def sample_negative_item(self, user):
        ix, rng = self.ix, self.rng
        if ix._neg_ptr is not None:
            cand = ix.negative_items(user)
            return int(cand[rng.randrange(len(cand))]) if len(cand) else None
        # rejection sampling; one extra draw always follows an accepted sample (utils.py:110-116)
        s = rng.randint(0, ix.total_items - 1)
        while True:
            ok = (not ix.has_positives(user)) or (not ix.is_positive(user, s))
            nxt = rng.randint(0, ix.total_items - 1)
            if ok:
                return s
            s = nxt

    def pairwise(self):
        while True:
            e = self.next_random_record()
            u = e["user_id"]
            yield u, e["item_id"], self.sample_negative_item(int(u))

    def stratified_pointwise(self, pos_ratio):
        ix, rng = self.ix, self.rng
        while True:
            if rng.random() <= pos_ratio:
                e = self.next_random_record()
                yield e["user_id"], e["item_id"], 1.0
            else:
                u = rng.randint(0, ix.total_users - 1); i = rng.randint(0, ix.total_items - 1)
                while ix.is_positive(u, i):
                    u = rng.randint(0, ix.total_users - 1); i = rng.randint(0, ix.total_items - 1)
                yield u, i, 0.0

    def per_pos_stratified_pointwise(self, pos_ratio):
        k = int((1 - pos_ratio) / pos_ratio)
        while True:
            e = self.next_random_record()
            u, p = e["user_id"], e["item_id"]
            yield u, p, 1.0
            count = 0
            for n in self.rng.sample(range(self.ix.total_items), k=k + 1):
                if n == p:
                    continue
                yield u, n, 0.0
                count += 1
                if count >= k:
                    break


class _BatchIterator:
    """Batches of a (possibly endless) sample stream; `take` bounds the number of batches."""

    def __init__(self, streams, keys, dtypes, batch_size, take):
        self.keys, self.dtypes, self.bs, self.take, self.count = keys, dtypes, batch_size, take, 0
        if len(streams) == 1:
            self._gen, self._q = streams[0], None
        else:                                           # independent producers, batches in arrival order
            self._q = queue.Queue(maxsize=len(streams))
            for g in streams:
                threading.Thread(target=self._produce, args=(g,), daemon=True).start()

    def _make(self, gen):
        cols = [[] for _ in self.keys]
        n = 0
        for sample in gen:
            for c, v in zip(cols, sample):
                c.append(v)
            n += 1
            if n == self.bs:
                break
        if n == 0:
            return None
        return {k: d(c) if callable(d) and not isinstance(d, type) else np.asarray(c, dtype=d) for k, c, d in zip(self.keys, cols, self.dtypes)}

    def _produce(self, gen):
        while True:
            b = self._make(gen)
            self._q.put(b)
            if b is None:
                return

    def __iter__(self):
        return self

    def __next__(self):
        if self.take is not None and self.count >= self.take:
            raise StopIteration
        b = self._make(self._gen) if self._q is None else self._q.get()
        if b is None:
            raise StopIteration
        self.count += 1
        return b


class Dataset:
    """`openrec.tf2.data.Dataset` surface (dataset.py:89-176)."""

    def __init__(self, raw_data, total_users, total_items, implicit_negative=True, num_negatives=None,
                 seed=None, sortby=None, asc=True, name=None):
        self.name = name
        random.seed(seed)                     # the reference seeds the module-global stream (utils.py:12)
        self.datastore = InteractionIndex(raw_data, total_users, total_items, implicit_negative, num_negatives,
                                          sortby, asc)
        self._main = _Sampler(self.datastore, random)       # in-process stream = the global `random` module

    def _streams(self, make, num_parallel_calls):
        if num_parallel_calls <= 1:
            return [make(self._main)]
        return [make(_Sampler(self.datastore, random.Random())) for _ in range(num_parallel_calls)]

    def pairwise(self, batch_size, num_parallel_calls=1, take=None):
        return _BatchIterator(self._streams(lambda s: s.pairwise(), num_parallel_calls),
                              ("user_id", "p_item_id", "n_item_id"), (np.int32,) * 3, batch_size, take)

    def stratified_pointwise(self, batch_size, pos_ratio=0.5, num_parallel_calls=1, take=None):
        return _BatchIterator(self._streams(lambda s: s.stratified_pointwise(pos_ratio), num_parallel_calls),
                              ("user_id", "item_id", "label"), (np.int32, np.int32, np.float32), batch_size, take)

    def per_pos_stratified_pointwise(self, batch_size, pos_ratio=0.5, num_parallel_calls=1, take=None):
        return _BatchIterator(self._streams(lambda s: s.per_pos_stratified_pointwise(pos_ratio), num_parallel_calls),
                              ("user_id", "item_id", "label"), (np.int32, np.int32, np.float32), batch_size, take)

    def evaluation(self, batch_size, excl_datasets=[]):
        """dataset.py:60-82 / :161-176.  The masks of a batch are `SparseMask` objects (one sorted item list per user -- what
        the index holds; the metrics take them as lists and `np.asarray(mask)` gives the reference's dense bool rows).  With
        explicit negatives the exclusion mask is "everything but the labelled items" and stays a dense array."""
        from ...runtime import SparseMask
        ix = self.datastore
        dense_excl = ix.contain_negatives()

        def gen():
            for u in ix.warm_users():
                pos = np.unique(np.asarray(ix.positive_items(u), np.int64))
                if dense_excl:
                    excl = np.ones(ix.total_items, bool)
                    excl[pos] = False
                    excl[ix.negative_items(u)] = False
                    for d in excl_datasets:
                        excl[d.datastore.positive_items(u)] = True
                else:
                    excl = [x for d in excl_datasets for x in d.datastore.positive_items(u)]
                yield u, pos, excl

        def lists(rows):
            return SparseMask.from_lists(rows, ix.total_items)
        return _BatchIterator([gen()], ("user_id", "pos_mask", "excl_mask"),
                              (np.int32, lists, bool if dense_excl else lists), batch_size, None)
