"""TEST INFRASTRUCTURE ONLY — ctypes wrapper of oracle/glove_ref.c (scalar fp32 CPU port).

Used by tests (C port vs numpy restatement) and by bench.py's `cpu_baseline` leg.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libglove_ref.so"


class State(C.Structure):
    _fields_ = [("V", C.c_int32), ("d", C.c_int32)] + [(n, C.c_void_p) for n in (
        "R", "C", "br", "bc", "S1_R", "S1_C", "S1_br", "S1_bc", "S2_R", "S2_C", "S2_br", "S2_bc", "scal")] + [
        ("step", C.c_int64)] + [(n, C.c_void_p) for n in (
            "G_R", "G_C", "G_br", "G_bc", "touch_r", "touch_c", "mark_r", "mark_c")]


class SideIndexC(C.Structure):
    _fields_ = [("n_uniq", C.c_int32), ("n_chunks", C.c_int32)] + [(n, C.c_void_p) for n in (
        "order", "uid", "first", "ch_lo", "ch_q")]


class IndexC(C.Structure):
    _fields_ = [("r", SideIndexC), ("c", SideIndexC)] + [(n, C.c_void_p) for n in ("e", "P_r", "P_c", "Pb_r", "Pb_c")]


class BatchIndex:
    """Grouping of one batch's pairs by row id and by col id for the all-core step (built once per resident
    batch, outside any timed region): stable order inside an id, ids with more than `chunk` pairs cut into chunks."""

    def __init__(self, row, col, d, chunk=256):
        self.keep = []

        def side(ids):
            order = np.argsort(ids, kind="stable").astype(np.int32)
            uid, start = np.unique(ids[order], return_index=True)
            cnt = np.diff(np.r_[start, len(ids)])
            nch = (cnt + chunk - 1) // chunk
            first = np.r_[0, np.cumsum(nch)].astype(np.int32)
            ch_q = np.repeat(np.arange(len(uid)), nch).astype(np.int32)
            within = np.arange(int(first[-1])) - first[:-1][ch_q]
            ch_lo = np.r_[start[ch_q] + within * chunk, len(ids)].astype(np.int32)
            arrs = [order, uid.astype(np.int32), first, ch_lo, ch_q]
            self.keep += arrs
            sx = SideIndexC()
            sx.n_uniq, sx.n_chunks = len(uid), int(first[-1])
            for n, a in zip(("order", "uid", "first", "ch_lo", "ch_q"), arrs):
                setattr(sx, n, np.ascontiguousarray(a).ctypes.data)
            return sx
        ix = IndexC()
        ix.r, ix.c = side(np.asarray(row)), side(np.asarray(col))
        self.e = np.zeros(len(row), np.float32)
        self.P_r, self.P_c = np.zeros((ix.r.n_chunks, d), np.float32), np.zeros((ix.c.n_chunks, d), np.float32)
        self.Pb_r, self.Pb_c = np.zeros(ix.r.n_chunks, np.float32), np.zeros(ix.c.n_chunks, np.float32)
        for n in ("e", "P_r", "P_c", "Pb_r", "Pb_c"):
            setattr(ix, n, getattr(self, n).ctypes.data)
        self.c_struct = ix


class HyperC(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("l2_reg", "reg_mult", "lr", "eps", "beta1", "beta2", "inv_batch")]


def build():
    subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    return LIB


class CPort:
    """fp32 tables + scratch living in numpy arrays, stepped by the C restatement."""

    def __init__(self, tables, max_batch):
        """`tables`: oracle.glove_ref.Tables (any dtype; copied to fp32)."""
        if not LIB.exists():
            build()
        self.lib = C.CDLL(str(LIB))
        t = tables.astype(np.float32)
        self.t = t
        V, d = t.V, t.d
        self.optimizer = t.optimizer
        s1 = "A_" if t.optimizer == "Adagrad" else "M_"
        self.arr = dict(R=t.R, C=t.C, br=t.br, bc=t.bc)
        for n in ("R", "C", "br", "bc"):
            self.arr["S1_" + n] = getattr(t, s1 + n)
            self.arr["S2_" + n] = getattr(t, "V_" + n) if t.optimizer == "Adam" else np.zeros(1, np.float32)
        scal = np.zeros(3, np.float32)
        scal[0] = t.g
        scal[1] = t.A_g if t.optimizer == "Adagrad" else t.M_g
        scal[2] = 0.0 if t.optimizer == "Adagrad" else t.V_g
        self.arr["scal"] = scal
        self.arr.update(G_R=np.zeros((V, d), np.float32), G_C=np.zeros((V, d), np.float32),
                        G_br=np.zeros(V, np.float32), G_bc=np.zeros(V, np.float32),
                        touch_r=np.zeros(max_batch, np.int32), touch_c=np.zeros(max_batch, np.int32),
                        mark_r=np.zeros(V, np.uint8), mark_c=np.zeros(V, np.uint8))
        st = State()
        st.V, st.d, st.step = V, d, t.step
        for k, a in self.arr.items():
            assert a.flags.c_contiguous
            setattr(st, k, a.ctypes.data)
        self.st = st
        self.out = np.zeros(3, np.float32)

    def step(self, row, col, w, y, hp, inv_batch=None):
        if getattr(hp, "head", 0) != 0:
            raise NotImplementedError("the scalar C port restates the GloVe (regression) head only")
        B = len(row)
        h = HyperC(hp.l2_reg, hp.reg_mult, hp.learning_rate, hp.epsilon, hp.beta1, hp.beta2,
                   (1.0 / B) if inv_batch is None else inv_batch)
        fn = self.lib.glove_ref_step_adagrad_f32 if self.optimizer == "Adagrad" else self.lib.glove_ref_step_adam_f32
        row, col = np.ascontiguousarray(row, np.int32), np.ascontiguousarray(col, np.int32)
        w, y = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(y, np.float32)
        rc = fn(C.byref(self.st), C.byref(h), row.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                w.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(B),
                self.out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return tuple(float(x) for x in self.out)

    def step_mt(self, index: "BatchIndex", row, col, w, y, hp, threads=0, inv_batch=None):
        """The same step on `threads` cores (0 = all OpenMP offers): oracle/glove_ref.c glove_ref_step_mt_f32."""
        B = len(row)
        h = HyperC(hp.l2_reg, hp.reg_mult, hp.learning_rate, hp.epsilon, hp.beta1, hp.beta2,
                   (1.0 / B) if inv_batch is None else inv_batch)
        row, col = np.ascontiguousarray(row, np.int32), np.ascontiguousarray(col, np.int32)
        w, y = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(y, np.float32)
        rc = self.lib.glove_ref_step_mt_f32(C.byref(self.st), C.byref(h), C.byref(index.c_struct),
                                            row.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                            w.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(B),
                                            C.c_int(0 if self.optimizer == "Adagrad" else 1), C.c_int(int(threads)),
                                            self.out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return tuple(float(x) for x in self.out)

    def max_threads(self) -> int:
        return int(self.lib.glove_ref_max_threads())

    @property
    def g(self):
        return float(self.arr["scal"][0])
