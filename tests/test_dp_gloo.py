"""Data-parallel host logic on CPU: world_size 2 over gloo, kernels replaced by the oracle.

Property under test (DESIGN.md "Multi-GPU"): N ranks stepping on their own batches of B
nonzeros with one all-reduce of the flat dense-gradient buffer == one rank stepping on the
concatenated batch of N*B nonzeros."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent / "oracle"))
from helpers import free_port  # noqa: E402

WORLD = 2
B, V, D, STEPS = 96, 40, 8, 5


def _batches():
    sys.path.insert(0, str(HERE))
    from helpers import make_batch
    return [[make_batch(100 * s + r, B, V) for r in range(WORLD)] for s in range(STEPS)]


def _worker(rank, port, optimizer, out_dir, exchange="dense", extra=None):
    extra = extra or {}
    for p in (HERE.parent, HERE.parent / "oracle", HERE):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import glove_ref as ref
    from oracle_backend import OracleBackend, OracleTables
    from trainer.stepper import Stepper
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    tables = OracleTables(ref.Tables(V, D, optimizer, dtype=np.float64, seed=3))
    backend = OracleBackend()
    stepper = Stepper(backend, tables, dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05, optimizer=optimizer, **extra), B, WORLD, dist,
                      exchange=exchange)
    assert stepper.dense and abs(stepper.hyper["inv_batch"] - 1.0 / (WORLD * B)) < 1e-15
    plans = [backend.build_plan(*step_batches[rank], V, 32) for step_batches in _batches()]
    stepper.prepare(plans)                       # collective: the ranks agree on the exchange
    want_rows = exchange == "rows" or optimizer in Stepper.ROWS_ONLY      # the per-row optimizers always travel as lists
    assert stepper.rows == want_rows and [n for n, _ in stepper.phases()].count("all_gather") == int(stepper.rows)
    if exchange == "auto" and optimizer == "Adagrad":      # 2 x ~80 ids x (d+4) floats against 2 V (d+1): dense is the shorter payload here
        assert not stepper.rows
    for plan in plans:
        stepper.step(plan)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), R=tables.t.R, C=tables.t.C, br=tables.t.br,
             bc=tables.t.bc, g=tables.t.g, step=tables.t.step)
    dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,exchange,extra", [
    ("Adagrad", "dense", {}), ("Adam", "dense", {}), ("Adagrad", "rows", {}), ("Adagrad", "auto", {}),
    # the other names tf.keras.optimizers.get resolves (train_utils.py:13-16): per-row ones on the touched-rows exchange
    # (whatever was asked for), the dense-decay RMSprop on the all-reduce
    ("SGD", "auto", {}), ("SGD", "rows", dict(momentum=0.9, nesterov=True)), ("Adamax", "auto", {}), ("Adadelta", "auto", {}),
    ("Ftrl", "auto", {}), ("RMSprop", "auto", {}),
    # Nadam: m and v decay everywhere, the union of the ranks' touched rows moves — the lists are that union
    ("Nadam", "auto", {})])
def test_two_ranks_equal_one_rank_on_the_joint_batch(tmp_path, optimizer, exchange, extra):
    """Dense all-reduce and touched-rows all-gather: either way two ranks == one rank on the joint batch."""
    sys.path.insert(0, str(HERE.parent / "oracle"))
    import glove_ref as ref
    port = free_port()
    mp.spawn(_worker, args=(port, optimizer, str(tmp_path), exchange, extra), nprocs=WORLD, join=True)
    t = ref.Tables(V, D, optimizer, dtype=np.float64, seed=3)
    hp = ref.Hyper(learning_rate=0.05, **extra)
    for step_batches in _batches():
        joint = [np.concatenate([b[i] for b in step_batches]) for i in range(4)]
        ref.train_step(t, *joint, hp)
    ranks = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(WORLD)]
    for name in ("R", "C", "br", "bc", "g"):
        np.testing.assert_array_equal(ranks[0][name], ranks[1][name])           # replicas stay identical
        np.testing.assert_allclose(ranks[0][name], getattr(t, name), rtol=1e-10, atol=1e-13)
    assert int(ranks[0]["step"]) == STEPS


# ---- BASELINE config 5: row table sharded over the ranks, nonzeros routed to the owners of their rows
def _sharded_worker(rank, port, out_dir, exchange="dense"):
    for p in (HERE.parent, HERE.parent / "oracle", HERE):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import glove_ref as ref
    from oracle_backend import OracleBackend, OracleTables
    from trainer.stepper import RowShardedStepper, owned_rows, route_by_row_owner
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    full = ref.Tables(V, D, "Adagrad", dtype=np.float64, seed=3)
    assert owned_rows(V, WORLD, rank) == len(range(rank, V, WORLD))
    shard = full.copy()                                   # R / br / their accumulators: rows rank, rank+W, ...
    for n in ("R", "br", "A_R", "A_br"):
        setattr(shard, n, getattr(full, n)[rank::WORLD].copy())
    tables = OracleTables(shard)
    backend = OracleBackend()
    stepper = RowShardedStepper(backend, tables, dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05), B, WORLD, dist,
                                exchange=exchange)
    routed_sizes, plans = [], []
    for step_batches in _batches():
        mine = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in zip(("row", "col", "w", "y"), step_batches[rank])}
        routed = route_by_row_owner(mine, WORLD, rank, dist)
        routed_sizes.append(int(routed["row"].numel()))
        assert int(routed["row"].max()) < tables.V_row
        plans.append(backend.build_plan(routed["row"].numpy(), routed["col"].numpy(), routed["w"].numpy(),
                                        routed["y"].numpy(), V, 32))
    stepper.prepare(plans)
    assert stepper.rows == (exchange == "rows")
    for plan in plans:
        stepper.step(plan)
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), R=shard.R, br=shard.br, C=shard.C, bc=shard.bc, g=shard.g,
             sizes=np.asarray(routed_sizes))
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["dense", "rows"])
def test_row_sharded_step_equals_single_rank_on_the_joint_batch(tmp_path, exchange):
    sys.path.insert(0, str(HERE.parent / "oracle"))
    import glove_ref as ref
    port = free_port()
    mp.spawn(_sharded_worker, args=(port, str(tmp_path), exchange), nprocs=WORLD, join=True)
    t = ref.Tables(V, D, "Adagrad", dtype=np.float64, seed=3)
    hp = ref.Hyper(learning_rate=0.05)
    for step_batches in _batches():
        ref.train_step(t, *[np.concatenate([b[i] for b in step_batches]) for i in range(4)], hp)
    shards = [np.load(tmp_path / ("shard%d.npz" % r)) for r in range(WORLD)]
    assert sum(int(s["sizes"].sum()) for s in shards) == STEPS * WORLD * B      # every nonzero routed exactly once
    for r, s in enumerate(shards):
        np.testing.assert_allclose(s["R"], t.R[r::WORLD], rtol=1e-10, atol=1e-13)    # each rank owns its rows
        np.testing.assert_allclose(s["br"], t.br[r::WORLD], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(s["C"], t.C, rtol=1e-10, atol=1e-13)              # replicas agree with the oracle
        np.testing.assert_allclose(s["bc"], t.bc, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(s["g"], t.g, rtol=1e-10)
    np.testing.assert_array_equal(shards[0]["C"], shards[1]["C"])


# ---- BASELINE config 5, both tables sharded: col rows fetched from / returned to their owners by all-to-all
def _fully_sharded_batches():
    """The common batches, plus two steps whose col ids all belong to ONE owner (even ids: rank 0; then odd ids):
    the other rank serves nothing and receives no gradients in that step, and still takes part in every collective."""
    out = _batches()
    for parity in (0, 1):
        step = []
        for r in range(WORLD):
            row, col, w, y = _batches()[parity][r]
            col = (col // 2 * 2 + parity) % V
            clash = col == row
            col[clash] = (col[clash] + 2) % V
            step.append((row, col.astype(np.int32), w, y))
        out.append(step)
    return out


def _fully_sharded_worker(rank, port, out_dir):
    for p in (HERE.parent, HERE.parent / "oracle", HERE):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import glove_ref as ref
    from oracle_backend import OracleBackend, OracleTables
    from trainer.stepper import ShardedStepper, route_by_row_owner
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    full = ref.Tables(V, D, "Adagrad", dtype=np.float64, seed=3)
    shard = full.copy()
    for n in ("R", "br", "A_R", "A_br", "C", "bc", "A_C", "A_bc"):     # rows AND cols: id % world == rank, local index id // world
        setattr(shard, n, getattr(full, n)[rank::WORLD].copy())
    tables = OracleTables(shard)
    backend = OracleBackend()
    stepper = ShardedStepper(backend, tables, dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05), B, WORLD, rank, dist)
    handles = []
    for step_batches in _fully_sharded_batches():
        mine = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in zip(("row", "col", "w", "y"), step_batches[rank])}
        routed = route_by_row_owner(mine, WORLD, rank, dist)
        handles.append(stepper.add_batch(routed["row"], routed["col"], routed["w"], routed["y"], 32))
    for h in handles:
        stepper.step(h)
    np.savez(os.path.join(out_dir, "full%d.npz" % rank), R=shard.R, br=shard.br, C=shard.C, bc=shard.bc, g=shard.g,
             A_C=shard.A_C, step=shard.step)
    dist.destroy_process_group()


def test_fully_sharded_step_equals_single_rank_on_the_joint_batch(tmp_path):
    sys.path.insert(0, str(HERE.parent / "oracle"))
    import glove_ref as ref
    port = free_port()
    mp.spawn(_fully_sharded_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    t = ref.Tables(V, D, "Adagrad", dtype=np.float64, seed=3)
    hp = ref.Hyper(learning_rate=0.05)
    for step_batches in _fully_sharded_batches():
        ref.train_step(t, *[np.concatenate([b[i] for b in step_batches]) for i in range(4)], hp)
    shards = [np.load(tmp_path / ("full%d.npz" % r)) for r in range(WORLD)]
    for r, s in enumerate(shards):
        for n in ("R", "br", "C", "bc", "A_C"):
            np.testing.assert_allclose(s[n], getattr(t, n)[r::WORLD], rtol=1e-10, atol=1e-13, err_msg=n)
        np.testing.assert_allclose(s["g"], t.g, rtol=1e-10)
        assert int(s["step"]) == STEPS + 2


# ---- the input side on several ranks: every nonzero of the file belongs to exactly one rank
def _stream_worker(rank, port, out_dir, routed):
    for p in (HERE.parent, HERE.parent / "oracle", HERE):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    from oracle_backend import OracleBackend
    from trainer.data_utils import NonzeroStream
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    rng = np.random.default_rng(0)
    n = 1001                                                   # odd: the data-parallel split has a remainder
    row = (rng.zipf(1.3, n) % V).astype(np.int32)              # skewed: id % world ownership is unbalanced
    coo = dict(row=row, col=rng.integers(0, V, n).astype(np.int32), w=np.arange(n, dtype=np.float32),
               y=rng.normal(size=n).astype(np.float32))
    st = NonzeroStream(coo, 50, V, OracleBackend(), "cpu", rank=rank, world=WORLD, seed=7, route=dist if routed else None)
    seen = sum(int(b[0].numel()) for b in st.eval_batches())
    assert seen == st.nnz and len(st.plans) == st.nnz // 50
    np.savez(os.path.join(out_dir, "stream%d.npz" % rank), w=st.w.numpy(), row=st.row.numpy(), nnz=st.nnz)
    dist.destroy_process_group()


@pytest.mark.parametrize("routed", [False, True])
def test_every_nonzero_lands_on_exactly_one_rank(tmp_path, routed):
    """Data-parallel shards (a contiguous slice of one permutation each, the remainder spread over the first ranks) and
    row-owner routing (nothing truncated to the lightest rank's count): the ranks' streams partition the file."""
    port = free_port()
    mp.spawn(_stream_worker, args=(port, str(tmp_path), routed), nprocs=WORLD, join=True)
    parts = [np.load(tmp_path / ("stream%d.npz" % r)) for r in range(WORLD)]
    w = np.sort(np.concatenate([p["w"] for p in parts]))       # the weights are the nonzeros' serial numbers
    np.testing.assert_array_equal(w, np.arange(1001, dtype=np.float32))
    assert sum(int(p["nnz"]) for p in parts) == 1001
    if routed:
        assert int(parts[0]["nnz"]) != int(parts[1]["nnz"])    # unbalanced ownership, and still nothing dropped


# ---- reshuffled epochs (--epoch-shuffle full, the reference's make_csv_dataset(shuffle=True, num_epochs=None)) on several ranks
RB, RN, RSTEPS = 40, 403, 24            # 201 / 202 pairs per rank -> 5 batches per epoch: 24 steps cross four epoch boundaries


class _Recorder:
    """Wraps a kernel provider: remembers the batch of every index it builds (the order the steps consume them in)."""

    def __init__(self, inner):
        self.inner, self.seen = inner, []

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def build_plan(self, row, col, w, y, V, chunk_cap):
        self.seen.append(tuple(np.array(a) for a in (row, col, w, y)))
        return self.inner.build_plan(row, col, w, y, V, chunk_cap)


def _reshuffle_worker(rank, port, out_dir, form):
    for p in (HERE.parent, HERE.parent / "oracle", HERE):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import glove_ref as ref
    from oracle_backend import OracleBackend, OracleTables
    from trainer.data_utils import NonzeroStream
    from trainer.stepper import ReshufflingRunner, RowShardedStepper, ShardedStepper, Stepper
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    rng = np.random.default_rng(0)
    coo = dict(row=rng.integers(0, V, RN).astype(np.int32), col=rng.integers(0, V, RN).astype(np.int32),
               w=np.arange(RN, dtype=np.float32) + 1.0, y=rng.normal(size=RN).astype(np.float32))    # w = the pair's serial number
    full = ref.Tables(V, D, "Adagrad", dtype=np.float64, seed=3)
    shard = full.copy()
    owner_major = form == "sharded_owner_major"     # col ids renumbered owner-major by the stream (NonzeroStream(cols_by_owner=))
    form = "sharded" if owner_major else form
    sharded_names = {"dp_dense": (), "dp_rows": (), "rowsharded": ("R", "br", "A_R", "A_br"),
                     "sharded": ("R", "br", "A_R", "A_br", "C", "bc", "A_C", "A_bc")}[form]
    for n in sharded_names:
        setattr(shard, n, getattr(full, n)[rank::WORLD].copy())
    tables = OracleTables(shard)
    backend = _Recorder(OracleBackend())
    stream = NonzeroStream(coo, RB, V, backend, "cpu", rank=rank, world=WORLD, seed=11, static_plans=False,
                           route=dist if sharded_names else None, cols_by_owner=WORLD if owner_major else 0)
    if owner_major:
        per = (V + WORLD - 1) // WORLD
        assert stream.col_per == per and stream.V_cols == WORLD * per and int(stream.col.max()) < WORLD * per
        back = (stream.col % per) * WORLD + stream.col // per           # the numbering is a bijection of the ids
        assert int(back.max()) < V
    kw = dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05)
    if form == "sharded":
        stepper = ShardedStepper(backend, tables, kw, RB, WORLD, rank, dist)
    elif form == "rowsharded":
        stepper = RowShardedStepper(backend, tables, kw, RB, WORLD, dist, exchange="rows")
        stepper.prepare(batch_size=RB)
    else:
        stepper = Stepper(backend, tables, kw, RB, WORLD, dist, exchange="rows" if form == "dp_rows" else "dense")
        stepper.prepare(batch_size=RB)
        assert stepper.rows == (form == "dp_rows")
    # routed streams differ in length between the ranks: the ranks agree on the number of steps, each cycles through its own epochs
    runner = ReshufflingRunner(None, stream, tables, stepper.hyper, chunk_cap=8, burst=7, stepper=stepper)
    done = 0
    while done < RSTEPS:
        done += runner.run(min(3, RSTEPS - done))         # bursts end at epoch / burst boundaries: all of them get crossed
    # (both tables sharded: the epoch's batches are prepared together with renumbered col ids: no per-step record)
    seen = backend.seen[-RSTEPS:] if form != "sharded" else None
    np.savez(os.path.join(out_dir, "re%d.npz" % rank), R=shard.R, C=shard.C, br=shard.br, bc=shard.bc, g=shard.g, step=shard.step,
             nnz=stream.nnz, bpe=runner.nb,
             **({"b%d_%d" % (i, j): a for i, bt in enumerate(seen) for j, a in enumerate(bt)} if seen is not None else {}))
    dist.destroy_process_group()


@pytest.mark.parametrize("form", ["dp_dense", "dp_rows", "rowsharded"])
def test_reshuffled_epochs_on_two_ranks_equal_one_rank_on_the_joint_stream(tmp_path, form):
    """Every rank re-permutes ITS shard each epoch and indexes each batch when it is used; per step the ranks' batches
    together are one global batch: two ranks == the oracle stepping on the joint stream, step by step.  Every epoch of a
    rank visits each of its pairs at most once (the `nnz mod B` behind the last full batch wait for the next permutation)
    and the batches of two epochs differ."""
    sys.path.insert(0, str(HERE.parent / "oracle"))
    import glove_ref as ref
    port = free_port()
    mp.spawn(_reshuffle_worker, args=(port, str(tmp_path), form), nprocs=WORLD, join=True)
    ranks = [np.load(tmp_path / ("re%d.npz" % r)) for r in range(WORLD)]
    t = ref.Tables(V, D, "Adagrad", dtype=np.float64, seed=3)
    hp = ref.Hyper(learning_rate=0.05)
    for s in range(RSTEPS):
        parts = []
        for r, rk in enumerate(ranks):
            row, col, w, y = (rk["b%d_%d" % (s, j)] for j in range(4))
            if form == "rowsharded":
                row = row * WORLD + r                      # the routed stream carries shard-local row ids
            parts.append((row, col, w, y))
        ref.train_step(t, *[np.concatenate([p[i] for p in parts]) for i in range(4)], hp)
    for r, rk in enumerate(ranks):
        sl = slice(r, None, WORLD) if form == "rowsharded" else slice(None)
        np.testing.assert_allclose(rk["R"], t.R[sl], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(rk["br"], t.br[sl], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(rk["C"], t.C, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(rk["g"], t.g, rtol=1e-10)
        assert int(rk["step"]) == RSTEPS
        # epochs: bpe batches each; within an epoch no pair twice, and the first batches of two epochs differ
        bpe = int(rk["bpe"])
        for e in range(RSTEPS // bpe):
            serial = np.concatenate([rk["b%d_2" % s] for s in range(e * bpe, (e + 1) * bpe)])
            assert len(np.unique(serial)) == len(serial) == bpe * RB
        assert not np.array_equal(np.sort(rk["b0_2"]), np.sort(rk["b%d_2" % bpe]))
    if form != "rowsharded":
        np.testing.assert_array_equal(ranks[0]["C"], ranks[1]["C"])
        np.testing.assert_array_equal(ranks[0]["R"], ranks[1]["R"])


def test_reshuffled_epochs_with_both_tables_sharded(tmp_path):
    """ShardedStepper under the reshuffling runner: the epoch's batches (fetch lists, renumbered col ids) are prepared
    collectively when the epoch starts.  The model after 24 steps over four epoch boundaries is finite, every rank made
    the same number of steps, re-running gives the same bits (the permutations are seeded), and so does a stream whose col
    ids are numbered owner-major."""
    port = free_port()
    outs = []
    for attempt, form in enumerate(("sharded", "sharded", "sharded_owner_major")):
        d = tmp_path / ("run%d" % attempt)
        d.mkdir()
        mp.spawn(_reshuffle_worker, args=(port + attempt, str(d), form), nprocs=WORLD, join=True)
        outs.append([np.load(d / ("re%d.npz" % r)) for r in range(WORLD)])
    for r in range(WORLD):
        for n in ("R", "C", "br", "bc", "g"):
            assert np.isfinite(outs[0][r][n]).all()
            np.testing.assert_array_equal(outs[0][r][n], outs[1][r][n])
            # col ids renumbered owner-major by the stream (the form the trainer's --shard-cols runs): the same owners, the same
            # fetch order, the same compact ids — the same bits
            np.testing.assert_array_equal(outs[0][r][n], outs[2][r][n])
        assert int(outs[0][r]["step"]) == RSTEPS
