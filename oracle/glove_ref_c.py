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

    @property
    def g(self):
        return float(self.arr["scal"][0])
