"""Input side of the training path.

The reference streams `interaction.csv` through `tf.data.experimental.make_csv_dataset`
(reference src/models/data_utils.py:4-26: four selected columns, string tokens, shuffle buffer,
batches of --batch-size) and maps tokens to ids with a StaticHashTable inside the graph
(reference src/models/model_utils.py:121-127; id = line number of vocab.txt, OOV -> 0).

Here the CSV is parsed ONCE into a binary COO of (row_id i32, col_id i32, weight f32, value f32)
that stays resident in HBM; the per-step input is a slice of it plus its dedup index.
"""
from __future__ import annotations

import glob
import hashlib
import logging
import os

import numpy as np
import torch

logger = logging.getLogger(__name__)
CSV_SLAB_ROWS = 4_000_000


def file_lines(fname) -> int:
    """Reference src/models/utils.py:4-9."""
    i = -1
    with open(fname, encoding="utf8") as f:
        for i, _ in enumerate(f):
            pass
    return i + 1


def read_vocab(vocab_txt) -> list:
    """Tokens in id order.  `vocab.txt` is written as "\\n".join(tokens) (reference
    src/data/text8.py:150), i.e. without a trailing newline; a token may be any string,
    including "nan"/"null"/"na"."""
    with open(vocab_txt, encoding="utf8") as f:
        text = f.read()
    tokens = text.split("\n")
    if tokens and tokens[-1] == "" and text.endswith("\n"):
        tokens.pop()
    return tokens


def get_string_id_table(vocab_txt) -> dict:
    """token -> id (line number); look up with `.get(token, 0)`: OOV -> 0 (model_utils.py:121-127).
    A token listed twice keeps its LAST line, as TextFileInitializer would fail on duplicates
    this is only a tie-break for malformed files."""
    return {tok: i for i, tok in enumerate(read_vocab(vocab_txt))}


def get_id_string_table(vocab_txt) -> list:
    """id -> token; ids outside the table read "<UNK>" (model_utils.py:130-136)."""
    return read_vocab(vocab_txt)


def _cache_key(paths, vocab_txt, columns) -> str:
    h = hashlib.sha1()
    for p in list(paths) + [vocab_txt]:
        st = os.stat(p)
        h.update(("%s:%d:%d;" % (os.path.abspath(p), st.st_size, int(st.st_mtime))).encode())
    h.update("|".join(columns).encode())
    return h.hexdigest()[:16]


def load_interaction_csv(file_pattern, vocab_txt, row_name="row_token", col_name="col_token",
                         weight_name="glove_weight", target_name="glove_value", cache_dir=None):
    """CSV -> dict(row i32[nnz], col i32[nnz], w f32[nnz], y f32[nnz]) as numpy arrays.

    Tokens are read as plain strings with NA parsing OFF (text8 contains the words "nan", "null",
    "na": reference README.md:54).  Rows keep file order.  With `cache_dir` the parsed COO is
    stored as `interaction-<key>.coo.npz` and reused while the CSV / vocab files are unchanged.
    """
    import pandas as pd
    paths = sorted(glob.glob(file_pattern)) or [file_pattern]
    columns = [row_name, col_name, weight_name, target_name]
    cache = None
    if cache_dir:
        cache = os.path.join(cache_dir, "interaction-%s.coo.npz" % _cache_key(paths, vocab_txt, columns))
        if os.path.exists(cache):
            z = np.load(cache)
            return {k: z[k] for k in ("row", "col", "w", "y")}
    table = get_string_id_table(vocab_txt)
    parts = []
    for p in paths:
        # read in slabs: the token columns of a 200 M-row file (BASELINE config 4) do not fit in host memory as
        # Python strings, the 16 B/row binary COO does
        for df in pd.read_csv(p, usecols=columns, dtype={row_name: str, col_name: str, weight_name: np.float32,
                                                         target_name: np.float32},
                              keep_default_na=False, na_filter=False, chunksize=CSV_SLAB_ROWS):
            row = df[row_name].map(table).fillna(0).to_numpy(np.int32)   # OOV -> 0
            col = df[col_name].map(table).fillna(0).to_numpy(np.int32)
            parts.append((row, col, df[weight_name].to_numpy(np.float32), df[target_name].to_numpy(np.float32)))
    out = {k: np.concatenate([p[i] for p in parts]) for i, k in enumerate(("row", "col", "w", "y"))}
    if cache:
        np.savez(cache, **out)
    logger.info("loaded %d nonzeros from %s", len(out["row"]), paths)
    return out


class NonzeroStream:
    """The nonzero stream of one rank, resident on the device, cut into fixed batches.

    Training batches follow the reference's `num_epochs=None` dataset: full batches only, an
    endless stream (data_utils.py:12-21).  The pairs are permuted once (seeded; the reference's
    shuffle is an unseeded 10k-row buffer over an already hash-shuffled file), cut into
    floor(nnz / B) batches whose dedup index is built once, and every epoch visits the batches
    in a fresh random order.  The `nnz mod B` pairs behind the last full batch are only used when
    the stream is re-cut (`recut()`), which draws a new permutation.
    """

    def __init__(self, coo: dict, batch_size: int, V: int, backend, device, rank=0, world=1, seed=None,
                 chunk_cap=0, static_plans=True, route=None, cols_by_owner=0, presharded=False):
        """`static_plans=False`: no index is built here; the caller re-permutes the pairs every epoch
        (`reshuffle_in_place`) and indexes each batch when it is used (--epoch-shuffle full).
        `cols_by_owner` = W > 0 (both tables sharded over W ranks): col ids are renumbered owner-major — id v becomes
        (v % W) * ceil(V / W) + v // W — so that ascending col order IS the order in which a batch's col rows are fetched
        from their owners (owner after owner, local index ascending): a dealt batch then arrives sorted for the sharded step
        as well (`col_per` = ceil(V / W), `V_cols` = W * col_per: the range of the renumbered ids)."""
        self.B, self.V, self.backend, self.device = int(batch_size), int(V), backend, torch.device(device)
        self.chunk_cap = chunk_cap
        self.gen = torch.Generator(device="cpu")
        if seed is None:
            self.gen.seed()
        else:
            self.gen.manual_seed(int(seed))
        # contiguous shard of a global permutation for this rank (DESIGN.md "Multi-GPU")
        n = len(coo["row"])
        perm = torch.randperm(n, generator=self.gen)
        # every nonzero belongs to exactly one rank: the first n % world ranks take one more
        lo, hi = rank * (n // world) + min(rank, n % world), (rank + 1) * (n // world) + min(rank + 1, n % world)
        mine = perm[lo:hi] if world > 1 and not presharded else perm
        def take(a):        # numpy arrays (the parsed CSV) or tensors already on the device (bench.py's synthetic workloads)
            if torch.is_tensor(a):
                return a[mine.to(a.device)].to(self.device)
            return torch.from_numpy(np.ascontiguousarray(a))[mine].to(self.device)
        self.row, self.col, self.w, self.y = take(coo["row"]), take(coo["col"]), take(coo["w"]), take(coo["y"])
        if route is not None:
            # row-sharded model: every nonzero moves to the rank that owns its row (one all-to-all at load), row ids
            # become local.  Every rank keeps ALL it received: the stream is endless (data_utils.py:12-21 num_epochs=None),
            # each rank cycles through its own batches and the ranks simply run the same number of steps; a rank that
            # owns heavier rows (ownership is id % world over Zipf-distributed ids) has more batches and revisits each
            # of them less often, which the log line below quantifies.
            from trainer.stepper import route_by_row_owner
            dist = route
            got = route_by_row_owner(dict(row=self.row, col=self.col, w=self.w, y=self.y), world, rank, dist)
            self.row, self.col, self.w, self.y = (got[k].contiguous() for k in ("row", "col", "w", "y"))
            have = torch.tensor([self.row.numel(), -self.row.numel()], dtype=torch.int64, device=self.device)
            dist.all_reduce(have, op=dist.ReduceOp.MAX)
            most, least = int(have[0].item()), -int(have[1].item())
            self.load_imbalance = most / max(least, 1)
            (logger.warning if self.load_imbalance > 1.5 else logger.info)(
                "row-sharded stream: %d nonzeros on this rank, most / least over the ranks = %d / %d (%.2fx): "
                "the nonzeros of the lighter ranks are revisited that much more often", self.row.numel(), most, least,
                self.load_imbalance)
        self.col_per, self.V_cols = 0, self.V
        if cols_by_owner:
            W = int(cols_by_owner)
            self.col_per = (self.V + W - 1) // W
            self.V_cols = W * self.col_per
            c = self.col.long()
            c = torch.where((c < 0) | (c >= self.V), torch.zeros_like(c), c)     # (an id outside the vocabulary is the unknown token, id 0)
            self.col = ((c % W) * self.col_per + c // W).to(torch.int32)
        self.nnz = int(self.row.numel())
        if self.nnz < self.B:
            raise ValueError("batch size %d exceeds the %d nonzeros of this rank" % (self.B, self.nnz))
        self.plans = []
        if static_plans:
            self.recut(first=True)
        self._order, self._pos = None, 0
        self._dev_gen = None
        # --epoch-shuffle full on the HIP backend: the pairs are sorted ONCE, into the two master orders (row-major and
        # col-major), and every epoch is DEALT from them — a keyed bijection seats the pairs, one stable counting-sort pass
        # per order by batch number writes the epoch so that every batch arrives sorted by row id and by col id
        # (glove_masters_build / glove_epoch_deal): no sort per batch.  `row`, `col`, `w`, `y` become the row-major order
        # (the eval pass reads them), `batch(b)` the row side of batch b of the current epoch.
        self.masters, self.epoch = None, -1
        hip = getattr(backend, "hip", None)
        if not static_plans and hip is not None and self.device.type == "cuda":
            self.masters = hip.build_masters(self.row.contiguous(), self.col.contiguous(), self.w.contiguous(), self.y.contiguous(),
                                             self.V_cols, getattr(backend, "shard_rows", 0) or 0)
            self.row, self.col, self.w, self.y = self.masters.row_major.arrays()
            from trainer.hip_api import Pairs
            self._sets = [(Pairs(self.nnz, self.device), Pairs(self.nnz, self.device)) for _ in range(2)]
            self._deal_ws = hip.deal_workspace(self.nnz, self.B, self.device)
            self.side = torch.cuda.Stream(device=self.device)       # deals — and the index builds that read them (trainer.stepper)
            self._dealt_upto, self._dealt_ev = -1, {}
            # whether work on the COMPUTE stream reads the epoch buffers (`batch(b)` handed to kernels there): a deal then
            # waits for what that stream holds before it overwrites a buffer set.  The runner, whose index builds run on
            # `side` itself, turns it off: its deals start at once
            self.main_reads_epochs = True

    # ---- dealt epochs
    def _ensure_dealt(self, e: int):
        """Epochs up to `e` have been dealt (issued on the side stream; epoch e lives in buffer set e % 2)."""
        hip = self.backend.hip
        while self._dealt_upto < e:
            nxt = self._dealt_upto + 1
            # the set's earlier readers: the index builds of epoch nxt - 2 (same stream) and whatever the compute stream
            # has been handed of it
            if self.main_reads_epochs:
                self.side.wait_stream(torch.cuda.current_stream(self.device))
            key = int.from_bytes(torch.randint(0, 256, (16,), generator=self.gen, dtype=torch.uint8).numpy().tobytes(), "little")
            with torch.cuda.stream(self.side):
                rs, cs = self._sets[nxt % 2]
                hip.deal_epoch(self.masters, self.B, key, rs, cs, self._deal_ws)
                ev = torch.cuda.Event()
                ev.record(self.side)
            self._dealt_ev[nxt] = ev
            self._dealt_ev.pop(nxt - 2, None)
            self._dealt_upto = nxt

    def epoch_sides(self, e: int | None = None):
        """(row side, col side) of epoch e (default: the current one) as trainer.hip_api.Pairs: batch k = positions
        [k B, (k + 1) B) of both, sorted by row id / by col id.  Valid on the side stream, or behind `dealt_event(e)`."""
        e = self.epoch if e is None else e
        self._ensure_dealt(e)
        return self._sets[e % 2]

    def dealt_event(self, e: int):
        self._ensure_dealt(e)
        return self._dealt_ev[e]

    def reshuffle_in_place(self):
        """The next epoch begins: a fresh permutation of this rank's pairs.

        Dealt stream (HIP): epoch e + 1 was dealt on the side stream while epoch e trained; the boundary is a wait on its
        event and the deal of epoch e + 2 is issued behind it.
        Otherwise (a CPU test backend): `row`, `col`, `w`, `y` are re-permuted; on a CUDA device without the library the next
        epoch's permutation is drawn and applied on a side stream into a second set of buffers."""
        if self.masters is not None:
            e = self.epoch + 1
            torch.cuda.current_stream(self.device).wait_event(self.dealt_event(e))
            self.epoch = e
            self._ensure_dealt(e + 1)
            return
        if self.device.type != "cuda":
            if self._dev_gen is None:
                self._dev_gen = torch.Generator(device=self.device)
                self._dev_gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,), generator=self.gen)))
            p = torch.randperm(self.nnz, generator=self._dev_gen, device=self.device)
            for t in (self.row, self.col, self.w, self.y):
                t.copy_(t[p])
            return
        if self._dev_gen is None:       # permutations drawn on the device (a host randperm of 10^6 pairs per epoch would
            self._dev_gen = torch.Generator(device=self.device)      # cost as much as the epoch's steps), seeded from the stream's generator
            self._dev_gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,), generator=self.gen)))
            self._spare = tuple(torch.empty_like(t) for t in (self.row, self.col, self.w, self.y))
            self._side = torch.cuda.Stream(device=self.device)
            self._ready = None
        main = torch.cuda.current_stream(self.device)
        if self._ready is None:
            self._permute_into_spare(main)                 # the first epoch: nothing was drawn ahead
        main.wait_event(self._ready)
        cur = (self.row, self.col, self.w, self.y)
        self.row, self.col, self.w, self.y = self._spare
        self._spare = cur
        self._permute_into_spare(main)                     # the next epoch's, behind everything that still reads the old buffers

    def _permute_into_spare(self, main):
        self._side.wait_stream(main)                       # the spare set's last readers (the epoch before) are on `main`
        with torch.cuda.stream(self._side):
            p = torch.randperm(self.nnz, generator=self._dev_gen, device=self.device)
            for dst, src in zip(self._spare, (self.row, self.col, self.w, self.y)):
                torch.index_select(src, 0, p, out=dst)
            self._ready = torch.cuda.Event()
            self._ready.record(self._side)

    def batch(self, b: int):
        """(row, col, w, y) of batch b of the current epoch — of a dealt epoch in row-major order (its arrival order)."""
        if self.masters is not None:
            if self.epoch < 0:
                raise RuntimeError("no epoch has begun: call reshuffle_in_place() first")
            return self._sets[self.epoch % 2][0].arrays(b * self.B, (b + 1) * self.B)
        s = slice(b * self.B, (b + 1) * self.B)
        return self.row[s], self.col[s], self.w[s], self.y[s]

    def recut(self, first=False):
        if not first:
            p = torch.randperm(self.nnz, generator=self.gen).to(self.device)
            self.row, self.col, self.w, self.y = self.row[p], self.col[p], self.w[p], self.y[p]
        nb = self.nnz // self.B
        self.plans = [self.backend.build_plan(*(t[b * self.B:(b + 1) * self.B] for t in
                                                (self.row, self.col, self.w, self.y)), self.V, self.chunk_cap)
                      for b in range(nb)]

    @property
    def batches_per_epoch(self) -> int:
        return self.nnz // self.B

    def next_plan(self):
        if self._order is None or self._pos >= len(self._order):
            self._order = torch.randperm(len(self.plans), generator=self.gen).tolist()
            self._pos = 0
        self.last_batch = self._order[self._pos]      # index b: pairs [b*B, (b+1)*B) of the stream
        plan = self.plans[self.last_batch]
        self._pos += 1
        return plan

    def eval_batches(self, batch_size=None):
        """One pass over every nonzero of this rank, final partial batch included (eval input_fn:
        num_epochs=1, estimator.py:87)."""
        bs = int(batch_size or self.B)
        for s in range(0, self.nnz, bs):
            yield self.row[s:s + bs], self.col[s:s + bs], self.w[s:s + bs], self.y[s:s + bs]
