"""ctypes binding of libglove_hip.so (C ABI: include/glove_hip.h).

This is the only way the host loop reaches the GPU kernels.  There is no CPU fallback: if the
shared library is missing or a call fails, an exception is raised.  torch is used here only
as the owner of device memory and streams (`tensor.data_ptr()`, `current_stream().cuda_stream`).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from pathlib import Path

import torch

logger = logging.getLogger(__name__)

PKG_DIR = Path(__file__).resolve().parent.parent
LIB_PATH = PKG_DIR / "lib" / "libglove_hip.so"

GLOVE_ABI_VERSION = 13
HEAD_REGRESSION, HEAD_LOGISTIC = 0, 1      # glove_hyper.head
OPTIMIZER_CODES = {"Adagrad": 0, "SGD": 1, "RMSprop": 2, "Adamax": 3, "Adam": 4, "Adadelta": 5, "Ftrl": 6, "Nadam": 7}      # glove_hyper.optimizer (GLOVE_OPT_*)
STEP_AUTO, STEP_TWO_LAUNCH, STEP_FUSED_ONE_PASS, STEP_FUSED_THREE_LAUNCH, STEP_FUSED_TWIN, STEP_TAGGED = 0, 1, 2, 3, 4, 5   # glove_hyper.step_form (2: tests / comparisons only)
TAGGED_STEP_MAX_BATCH = 2048      # GLOVE_STEP_AUTO takes the tagged step up to this batch size on step-tagged tables
DEFAULT_CHUNK_CAP = 32
RECORDS_AT_BUILD_MAX = 4096     # batches up to this size get their chunk records inside glove_plan_build
HEAVY_CHUNKS = 8          # ids with more chunks than this are reduced by a whole workgroup


RUN_WORDS_MIN_CHUNKS = 98304    # chunks of a side from which a resident plan of the fused regime keeps run words instead of records (6 chunks per lane group)
FUSED_STEP_BYTES = 96 << 20    # GLOVE_FUSED_STEP_BYTES (include/glove_hip.h): touched ids x row bytes x 4 beyond which the fused step pays (tests/test_abi.py holds the two together)


def auto_chunk_cap(B: int, V: int, d: int | None = None) -> int:
    """Chunk length used when the caller does not choose one.  Short chunks shorten the dependent
    chain of the gather passes (fewer partner-row round trips per chunk) and win while a batch holds
    few pairs per id; long chunks mean fewer partial rows and win for dense batches (bench.py --chunk-cap,
    V = 10000: B = 131072: 8 / 16 / 24 / 32 -> 22.4 / 21.2 / 22.3 / 23.3 us per step; B = 1048576: 16 -> 59.4, 32 -> 53.8).
    Tables beyond the L2s (`d` given: V x d x 4 B >= 32 MB) take the fused step at realistic batch sizes, which is bandwidth-bound
    and likes long chunks (V = 400 k, d = 300, B = 1 M: 8 / 16 / 32 -> 746 / 734 / 722 us per step)."""
    if d is not None and V * d * 4 >= (32 << 20):       # (round 5: V = 50 k, d = 300 — 61 MB — cap 8 / 16 / 32 -> 96.3 / 92.6 / 91.7 us per step)
        return 32
    return 16 if B <= 20 * V else 32

# every symbol include/glove_hip.h declares
EXPORTED_SYMBOLS = (
    "glove_abi_version", "glove_plan_workspace_bytes", "glove_plan_build", "glove_plan_fill_records", "glove_step_workspace_bytes",
    "glove_passes_f32", "glove_rowpass_f32", "glove_colpass_f32", "glove_apply_adagrad_f32",
    "glove_dense_grad_f32", "glove_dense_adagrad_f32", "glove_dense_adam_f32", "glove_step_adagrad_f32",
    "glove_steps_adagrad_f32", "glove_step_adam_f32", "glove_steps_adam_f32", "glove_eval_f32", "glove_eval_logistic_f32", "glove_topk_workspace_bytes", "glove_topk_cosine_f32",
    "glove_cooc_workspace_bytes", "glove_cooccurrence_i32", "glove_dense_grad_layout",
    "glove_pack_grad_f32", "glove_passes_packing_f32", "glove_pack_rest_f32", "glove_loss_partials_f32", "glove_combine_packed_f32", "glove_apply_packed_adagrad_f32",
    "glove_gather_rows_f32", "glove_canonicalize_f32", "glove_rowside_step_adagrad_f32",
    "glove_count_packed_f32",
    "glove_masters_workspace_bytes", "glove_masters_build", "glove_epoch_deal_workspace_bytes", "glove_epoch_deal",
    "glove_plan_sorted_workspace_bytes", "glove_plan_chunk_bound", "glove_plan_build_sorted", "glove_step_sparse_f32",
)

_fp = C.c_void_p  # device pointers travel as integers


class GloveTables(C.Structure):
    _fields_ = [("V", C.c_int32), ("d", C.c_int32), ("V_row", C.c_int32), ("d_model", C.c_int32),
                ("R", _fp), ("C", _fp), ("br", _fp), ("bc", _fp),
                ("s1_R", _fp), ("s1_C", _fp), ("s1_br", _fp), ("s1_bc", _fp),
                ("s2_R", _fp), ("s2_C", _fp), ("s2_br", _fp), ("s2_bc", _fp),
                ("scalars", _fp), ("step", _fp), ("R_ver", _fp), ("R_tag", _fp), ("C_tag", _fp)]


class GloveHyper(C.Structure):
    _fields_ = [("beta1", C.c_double), ("beta2", C.c_double),
                ("l2_reg", C.c_float), ("reg_mult", C.c_float), ("learning_rate", C.c_float),
                ("epsilon", C.c_float), ("inv_batch", C.c_float), ("sides", C.c_int32),
                ("head", C.c_int32), ("neg_factor", C.c_float), ("step_form", C.c_int32),
                ("optimizer", C.c_int32), ("momentum", C.c_float), ("nesterov", C.c_int32), ("rho", C.c_float)]


class GlovePlan(C.Structure):
    _fields_ = [("B", C.c_int64), ("chunk_cap", C.c_int32), ("cap_chunks", C.c_int32),
                ("cap_uniq", C.c_int32), ("heavy_chunks", C.c_int32), ("cap_heavy", C.c_int32),
                ("V_row", C.c_int32), ("counts", _fp), ("host_counts", C.c_int32 * 8),
                ("r_partner", _fp), ("r_w", _fp), ("r_y", _fp), ("r_to_c", _fp),
                ("r_chunk_id", _fp), ("r_chunk_start", _fp), ("r_uniq_slot", _fp), ("r_uniq_rec", _fp),
                ("c_partner", _fp), ("c_perm", _fp), ("c_w", _fp), ("c_y", _fp),
                ("c_chunk_id", _fp), ("c_chunk_start", _fp), ("c_uniq_slot", _fp), ("c_uniq_rec", _fp), ("heavy", _fp),
                ("r_crec", _fp), ("c_crec", _fp), ("r_mark", _fp), ("c_mark", _fp),
                ("r_chunk_hw", _fp), ("c_chunk_hw", _fp)]


class GlovePairs(C.Structure):
    _fields_ = [("id", _fp), ("partner", _fp), ("w", _fp), ("y", _fp)]


class GlovePackedList(C.Structure):
    _fields_ = [("entries", _fp), ("ids", _fp), ("header", _fp), ("n", C.c_int32), ("side", C.c_int32)]


class GloveHipError(RuntimeError):
    pass


_lib = None


def load_library(path: os.PathLike | None = None, any_abi: bool = False) -> C.CDLL:
    """dlopen libglove_hip.so and declare the prototypes.  Raises if it is not built.  `path` / `any_abi`: explicit
    arguments of the A/B tools (tools/ab_kernels.py), which load other builds of the library — older ones included —
    beside the shipped one; nothing in the product passes them and no environment variable is read."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise GloveHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    lib = C.CDLL(str(p))
    P = C.POINTER
    sz, i32, i64, vp = C.c_size_t, C.c_int32, C.c_int64, C.c_void_p
    protos = {
        "glove_abi_version": (C.c_int, []),
        "glove_plan_workspace_bytes": (sz, [i64, i32]),
        "glove_plan_build": (C.c_int, [vp, vp, vp, vp, i64, i32, P(GlovePlan), vp, sz, vp]),
        "glove_plan_fill_records": (C.c_int, [P(GlovePlan), vp]),
        "glove_masters_workspace_bytes": (sz, [i64]),
        "glove_masters_build": (C.c_int, [vp, vp, vp, vp, i64, i32, i32, P(GlovePairs), P(GlovePairs), vp, vp, vp, sz, vp]),
        "glove_epoch_deal_workspace_bytes": (sz, [i64, i64]),
        "glove_epoch_deal": (C.c_int, [P(GlovePairs), P(GlovePairs), vp, i64, i64, C.c_uint64, C.c_uint64, P(GlovePairs),
                                       P(GlovePairs), vp, sz, vp]),
        "glove_plan_sorted_workspace_bytes": (sz, [i64, i32]),
        "glove_plan_chunk_bound": (i32, [i64, i32, i32]),
        "glove_plan_build_sorted": (C.c_int, [P(GlovePairs), P(GlovePairs), i64, i64, i32, i32, P(GlovePlan), vp, vp, sz, vp]),
        "glove_step_workspace_bytes": (sz, [i64, i32, i32]),
        "glove_passes_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp]),
        "glove_rowpass_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp]),
        "glove_colpass_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp]),
        "glove_apply_adagrad_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, vp]),
        "glove_dense_grad_layout": (sz, [i32, i32, i32, vp]),
        "glove_dense_grad_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, vp]),
        "glove_dense_adagrad_f32": (C.c_int, [P(GloveTables), P(GloveHyper), vp, vp, vp]),
        "glove_dense_adam_f32": (C.c_int, [P(GloveTables), P(GloveHyper), vp, vp, vp]),
        "glove_step_adagrad_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, vp]),
        "glove_steps_adagrad_f32": (C.c_int, [P(P(GlovePlan)), i32, P(GloveTables), P(GloveHyper), vp, sz, vp, vp]),
        "glove_step_adam_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, vp, vp]),
        "glove_step_sparse_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, vp, vp]),
        "glove_steps_adam_f32": (C.c_int, [P(P(GlovePlan)), i32, P(GloveTables), P(GloveHyper), vp, sz, vp, vp, vp]),
        "glove_pack_grad_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, i64, vp]),
        "glove_passes_packing_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, i64, vp]),
        "glove_pack_rest_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp, i64, vp]),
        "glove_loss_partials_f32": (C.c_int, [P(GlovePlan), P(GloveTables), vp, sz, vp, vp]),
        "glove_combine_packed_f32": (C.c_int, [P(GlovePackedList), i32, P(GloveTables), vp, vp, i64, vp]),
        "glove_count_packed_f32": (C.c_int, [P(GlovePackedList), i32, P(GloveTables), vp, vp, i64, vp]),
        "glove_apply_packed_adagrad_f32": (C.c_int, [P(GlovePackedList), i32, P(GloveTables), P(GloveHyper), vp, vp, vp, vp, i64, vp]),
        "glove_gather_rows_f32": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp]),
        "glove_canonicalize_f32": (C.c_int, [P(GloveTables), vp]),
        "glove_rowside_step_adagrad_f32": (C.c_int, [P(GlovePlan), P(GloveTables), P(GloveHyper), vp, sz, vp]),
        "glove_eval_f32": (C.c_int, [vp, vp, vp, vp, i64, P(GloveTables), vp, vp]),
        "glove_eval_logistic_f32": (C.c_int, [vp, vp, vp, vp, i64, P(GloveTables), vp, vp]),
        "glove_topk_workspace_bytes": (sz, [i32, i32, i32]),
        "glove_topk_cosine_f32": (C.c_int, [vp, i32, i32, vp, i32, i32, vp, vp, vp, sz, vp]),
        "glove_cooc_workspace_bytes": (sz, [i64, i32]),
        "glove_cooccurrence_i32": (C.c_int, [vp, i64, i32, i32, vp, vp, vp, vp, vp, i64, vp, sz, vp]),
    }
    for name, (res, args) in protos.items():
        if path and any_abi and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.glove_abi_version() != GLOVE_ABI_VERSION and not (path and any_abi):
        raise GloveHipError("ABI mismatch: library %d, binding %d" % (lib.glove_abi_version(), GLOVE_ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        names = {-1: "GLOVE_E_BADARG", -2: "GLOVE_E_WORKSPACE"}
        raise GloveHipError("%s failed: %s" % (what, names.get(rc, "hipError %d" % rc)))


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GloveHipError("device tensor expected (the HIP path has no CPU fallback)")


def _require(t, dtype, n=None):
    """The kernels read raw pointers: a wrong dtype (torch's default int64 ids), a strided view or a short buffer
    would be silently misread."""
    if t is None:
        return
    if not t.is_cuda:
        raise GloveHipError("device tensor expected (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise GloveHipError("expected a %s tensor, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise GloveHipError("expected a contiguous tensor")
    if n is not None and t.numel() != n:
        raise GloveHipError("expected %d elements, got %d" % (n, t.numel()))


def row_width(rows: int, d: int) -> int:
    """Floats per stored table row (the row stride the kernels take) for an embedding size `d`: rows start on 16-byte
    boundaries at least; when the padding costs at most d / 12 floats they start on 64-byte boundaries, and on 128-byte lines
    for tables beyond the Infinity Cache (>= 128 MB: what the fused twin form also looks at).  d = 300: 304 floats (1,216 B =
    19 x 64) on a 60 MB table, 320 (1,280 B = 10 lines) on a 480 MB one; 64 and 128 stay as they are, 50 becomes 52.
    Measured on the same batches (tools/exp_row_stride.py, one process): V = 400 k, B = 1 M: 1,200-byte rows 622 - 628 us per step,
    1,216 B 608 - 609, 1,280 B 602 — 6.7 % more bytes per row and 3.7 % less time: rows that start mid-line cost two partial
    lines per access, on the write side two partial write-backs; V = 50 k (cache-resident): 101.5 / 99.8 - 100.4 / 101.7 - 103.0.
    The padding columns are zero and stay zero (glove_tables.d_model keeps lambda / d on the model's size)."""
    base = (int(d) + 3) // 4 * 4
    if os.environ.get("GLOVE_ROW_ALIGN") == "4":        # experiments (tools/): the former rule
        return base
    for align, ok in ((32, int(rows) * int(d) * 4 >= (128 << 20)), (16, True)):
        cand = (int(d) + align - 1) // align * align
        if ok and cand - int(d) <= int(d) // 12:
            return cand
    return base


class DeviceTables:
    """The five variables + optimizer slots as device buffers (reference model_utils.py:31-39)."""

    NAMES = ("R", "C", "br", "bc")

    def __init__(self, V: int, d: int, optimizer: str, device="cuda:0", seed: int | None = None,
                 V_row: int | None = None, V_col: int | None = None):
        """`d`: --embedding-size, any positive int.  Rows are stored aligned: `self.d` is the row stride
        (`row_width`: d rounded up to a multiple of 4, 16 or 32 floats — what the kernels and workspace queries take),
        `self.d_model` the reference's embedding size; the padding columns are zero and stay zero (glove_tables.d_model).
        `V_row` < V: this process holds only a shard of the row table (row ids handed to the kernels
        are then local indices into the shard).  `V_col` < V: it also holds only a shard of the col table
        (trainer.stepper.ShardedStepper, which runs the passes against fetched col rows); such tables go through
        TablesView, the plain kernels expect V col rows."""
        if d <= 0:
            raise ValueError("embedding size must be positive, got %d" % d)
        if optimizer not in OPTIMIZER_CODES:
            raise ValueError("optimizer must be one of %s (Keras names), got %r" % (", ".join(OPTIMIZER_CODES), optimizer))
        self.V, self.d_model, self.optimizer, self.device = int(V), int(d), optimizer, torch.device(device)
        self.V_row = int(V if V_row is None else V_row)
        self.V_col = int(V if V_col is None else V_col)
        self.d = row_width(self.V, d)        # (by the whole vocabulary, not this rank's shard: every rank takes the same stride)
        if not 0 < self.V_row <= self.V or not 0 < self.V_col <= self.V:
            raise ValueError("V_row and V_col must be in (0, V]")
        gen = torch.Generator(device="cpu")
        if seed is not None:
            gen.manual_seed(seed)
        else:
            gen.seed()   # the reference is unseeded (train_utils.py:26-27)

        def uni(*shape):  # Keras Embedding default: U(-0.05, 0.05)
            return ((torch.rand(*shape, generator=gen, dtype=torch.float32) - 0.5) * 0.1).to(self.device)

        def table(rows):
            t = torch.zeros(rows, self.d, dtype=torch.float32, device=self.device)
            t[:, :self.d_model] = uni(rows, self.d_model)
            return t

        self._R, self._C = table(self.V_row), table(self.V_col)
        self._br, self._bc = uni(self.V_row), uni(self.V_col)
        self.R_ver = None           # uint8[V_row] once enable_twin() has doubled R and br (glove_tables.R_ver)
        self.R_tag = self.C_tag = None    # once enable_tags() has doubled both tables (glove_tables.R_tag)
        self.scalars = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.s1, self.s2 = {}, {}
        for n in self.NAMES:
            w = getattr(self, n)
            if optimizer in ("Adagrad", "Ftrl"):
                self.s1[n] = torch.full_like(w, 0.1)   # initial_accumulator_value (both optimizers' Keras default)
            else:                                      # Adam m / v, Adamax m / v, Adadelta accum_grad / accum_var; SGD momentum accumulator, RMSprop rms: zeros
                self.s1[n] = torch.zeros_like(w)
            if optimizer in ("Adam", "Adamax", "Adadelta", "Ftrl", "Nadam"):
                self.s2[n] = torch.zeros_like(w)       # (Ftrl: linear)
        if optimizer in ("Adagrad", "Ftrl"):
            self.scalars[1] = 0.1
        if optimizer == "Nadam":
            self.scalars[4:6] = 1.0                    # the momentum cache (Keras: an optimizer weight initialised to ones)
        self._struct = None

    # ---- the row table may be twinned (glove_tables.R_ver): R / br are the plain views [V_row, ...]; reading them
    # (or handing the tables to anything but the Adagrad step) first brings the table back to its plain form
    @property
    def R(self) -> torch.Tensor:
        self.canonicalize()
        return self._R[:self.V_row]

    @R.setter
    def R(self, value):                     # `tables.R += x` and `tables.R = tensor` write into the buffer the kernels see
        view = self.R
        if value.data_ptr() != view.data_ptr():
            view.copy_(value)

    @property
    def br(self) -> torch.Tensor:
        self.canonicalize()
        return self._br[:self.V_row]

    @br.setter
    def br(self, value):
        view = self.br
        if value.data_ptr() != view.data_ptr():
            view.copy_(value)

    @property
    def C(self) -> torch.Tensor:
        self.canonicalize()
        return self._C[:self.V_col]

    @C.setter
    def C(self, value):
        view = self.C
        if value.data_ptr() != view.data_ptr():
            view.copy_(value)

    @property
    def bc(self) -> torch.Tensor:
        self.canonicalize()
        return self._bc[:self.V_col]

    @bc.setter
    def bc(self, value):
        view = self.bc
        if value.data_ptr() != view.data_ptr():
            view.copy_(value)

    def enable_tags(self):
        """Second copies of BOTH tables + a step tag per row (glove_tables.R_tag / C_tag): what the tagged step
        (GLOVE_STEP_TAGGED) needs to read pre-step rows while it updates rows in the same launch.  For the latency-bound
        regime — small tables: costs (V_row + V) x d x 4 B of HBM."""
        if self.R_tag is not None or self.optimizer not in ("Adagrad", "Adam") or self.R_ver is not None:
            return                               # (Adam: the twins flip as a whole every step — scalars[3] —, the tags stay zero)
        if self.V_col != self.V or 2 * max(self.V_row, self.V) * self.d * 4 >= 1 << 32:
            return                               # a sharded col table goes through views; 32-bit row offsets
        for name in ("_R", "_br", "_C", "_bc"):
            old = getattr(self, name)
            new = torch.zeros((2 * old.shape[0],) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
            new[:old.shape[0]].copy_(old)
            setattr(self, name, new)
        self.R_tag = torch.zeros(self.V_row, dtype=torch.int64, device=self.device)
        self.C_tag = torch.zeros(self.V, dtype=torch.int64, device=self.device)
        self._struct = None

    def maybe_enable_tags(self, batch_size: int):
        """The policy: batches the library steps in the tagged form (at most TAGGED_STEP_MAX_BATCH pairs) on tables small enough
        that the step is a latency chain (both tables within the caches: 64 MB)."""
        small = batch_size <= TAGGED_STEP_MAX_BATCH and (self.V_row + self.V) * self.d * 4 <= (64 << 20)
        if small and (self.optimizer == "Adagrad" or (self.optimizer == "Adam" and 2 * batch_size <= self.V_row + self.V)):
            self.enable_tags()

    def enable_twin(self):
        """Second copy of the row table + per-row version bytes: lets the fused step write a row's update beside the old
        row instead of through a partial-row slot (GLOVE_STEP_FUSED_TWIN).  Costs V_row x d x 4 B of HBM."""
        if self.R_ver is not None or self.optimizer != "Adagrad" or self.R_tag is not None:
            return
        if 2 * self.V_row * self.d * 4 >= 1 << 32:
            return                               # 32-bit row offsets: the table cannot be doubled
        for name in ("_R", "_br"):
            old = getattr(self, name)
            new = torch.zeros((2 * old.shape[0],) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
            new[:old.shape[0]].copy_(old)
            setattr(self, name, new)
        self.R_ver = torch.zeros(self.V_row, dtype=torch.uint8, device=self.device)
        self._struct = None

    def maybe_enable_twin(self):
        """The policy: row tables of 32 MB and more — where batches reach the fused step's regime — get the twin (128 MB until round 5).
        Measured per step, three-launch form -> twin form with its id triage: V = 400 k, d = 300: 678 -> 627 us;
        V = 2 M, d = 128: 578 -> 539 us (the passes pay ~20 us each for looking up which copy of a row is current, the
        apply launch shrinks from 95 to 15 us); V = 50 k, d = 300 (Infinity-Cache resident): no gain, not enabled."""
        # (round 4, tools/ab_step_forms.py: on the 61 MB table of V = 50 k, d = 300 — Infinity-Cache resident — the three-launch
        # form beats the twin form 101.0 to 103.2 us; on tables beyond the cache the twin form wins by 4 - 11 %)
        # (round 5: without a triage launch — the list of unfinished ids rides in the col-side launch — and without the streaming
        # cache policy on tables the caches hold, the twin form also wins on the 61 MB table: V = 50 k, d = 300, B = 131,072
        # 96.2 -> 92.9 us per step, B = 65,536 64.9 -> 64.4, B = 1 M 295.0 -> 292.6: tools/ab_step_forms.py,
        # profiles/r05_exp_twin_form_on_cache_resident_tables.txt; below 32 MB nothing was measured: not enabled)
        if self.optimizer == "Adagrad" and self.V_row * self.d * 4 >= (32 << 20):
            self.enable_twin()

    def canonicalize(self):
        """Versions back to 0 (a no-op without a twin; one sweep over V_row bytes plus the rows whose second copy was
        current).  Called by every accessor except the Adagrad step's."""
        # `_twin_dirty` is sticky: once a step that may flip versions has been issued (possibly inside a captured graph
        # that is replayed without any further Python call), every reader pays this one small launch
        if (self.R_ver is not None or self.R_tag is not None) and getattr(self, "_twin_dirty", False):
            _check(load_library().glove_canonicalize_f32(C.byref(self.struct(twin_ok=True)), _stream()),
                   "glove_canonicalize_f32")

    def struct(self, twin_ok=False) -> GloveTables:
        """`twin_ok`: the caller is the Adagrad step, which understands a twinned table (and says so through
        `_twin_dirty` when it may leave one behind: GloveHip.step_adagrad)."""
        if not twin_ok:
            self.canonicalize()
        if self._struct is None:
            s = GloveTables()
            s.V, s.d, s.V_row = self.V, self.d, (0 if self.V_row == self.V else self.V_row)
            s.d_model = 0 if self.d_model == self.d else self.d_model
            for n in self.NAMES:
                setattr(s, n, _ptr(getattr(self, "_" + n)))
                setattr(s, "s1_" + n, _ptr(self.s1[n]))
                setattr(s, "s2_" + n, _ptr(self.s2.get(n)))
            s.scalars, s.step, s.R_ver = _ptr(self.scalars), _ptr(self.step), _ptr(self.R_ver)
            s.R_tag, s.C_tag = _ptr(self.R_tag), _ptr(self.C_tag)
            self._struct = s
        return self._struct

    def embeddings(self, name: str) -> torch.Tensor:
        """R or C without the alignment padding: [rows, d_model] (a view)."""
        return getattr(self, name)[:, :self.d_model]

    # ---- (de)serialisation used by the checkpoint code and the tests: logical shapes [rows, d_model]
    def _logical(self, x):
        return (x[:, :self.d_model] if x.dim() == 2 else x).cpu()

    def _store(self, dst, src):
        if dst.dim() == 2:
            dst[:, :self.d_model].copy_(src)
        else:
            dst.copy_(src)

    def state_dict(self) -> dict:
        self.canonicalize()             # (also settles scalars[3], the current copy of tables the one-launch Adam step flips)
        out = {"V": self.V, "d": self.d_model, "V_row": self.V_row, "optimizer": self.optimizer,
               "scalars": self.scalars.cpu(), "global_step": self.step.cpu()}
        for n in self.NAMES:
            out[n] = self._logical(getattr(self, n))
            out["slot1_" + n] = self._logical(self.s1[n])
            if n in self.s2:
                out["slot2_" + n] = self._logical(self.s2[n])
        return out

    def load_state_dict(self, sd: dict):
        if (sd["V"], sd["d"], sd["optimizer"], sd.get("V_row", sd["V"])) != (self.V, self.d_model, self.optimizer, self.V_row):
            raise ValueError("checkpoint is for V=%s d=%s %s, model is V=%d d=%d %s" % (
                sd["V"], sd["d"], sd["optimizer"], self.V, self.d_model, self.optimizer))
        self.canonicalize()
        self.scalars.copy_(sd["scalars"])
        self.step.copy_(sd["global_step"])
        for n in self.NAMES:
            self._store(getattr(self, n), sd[n])
            self._store(self.s1[n], sd["slot1_" + n])
            if n in self.s2:
                self._store(self.s2[n], sd["slot2_" + n])

    # ---- row-sharded tables (BASELINE config 5): row u lives on rank u % world at local index u // world
    ROW_SIDE = ("R", "br")

    COL_SIDE = ("C", "bc")

    def _sharded_names(self):
        return self.ROW_SIDE + (self.COL_SIDE if self.V_col < self.V else ())

    def gather_whole(self, x, dist, world: int):
        """A table sharded by id % world (this rank's shard `x`) as the whole [V, ...] array, on every rank (collective)."""
        per = (self.V + world - 1) // world
        pad = torch.zeros((per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[:x.shape[0]] = x
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        return torch.stack(parts, 1).reshape((per * world,) + tuple(x.shape[1:]))[:self.V]

    def gather_by_owner(self, x, dist, world: int):
        """A table sharded by id % world as [world * ceil(V / world), ...]: rank after rank, each shard padded to the same
        length — row (v % world) * per + v // world holds id v (the numbering of NonzeroStream(cols_by_owner=world))."""
        per = (self.V + world - 1) // world
        pad = torch.zeros((per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[:x.shape[0]] = x
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        return torch.cat(parts, 0)

    def gathered_state_dict(self, dist, world: int) -> dict:
        """state_dict() of the WHOLE model: the shards of all ranks (the row side; the col side too when it is sharded) are
        all-gathered and interleaved back into [V, ...] arrays, so the checkpoint has the same format as an unsharded
        run's (collective)."""
        out = self.state_dict()
        out["V_row"] = self.V
        for n in self._sharded_names():
            out[n] = self._logical(self.gather_whole(getattr(self, n), dist, world))
            out["slot1_" + n] = self._logical(self.gather_whole(self.s1[n], dist, world))
            if n in self.s2:
                out["slot2_" + n] = self._logical(self.gather_whole(self.s2[n], dist, world))
        return out

    def load_whole_state_dict(self, sd: dict, world: int, rank: int):
        """Takes this rank's rows out of a whole-model state dict (the inverse of gathered_state_dict)."""
        if sd.get("V_row", sd["V"]) != sd["V"]:
            raise ValueError("the checkpoint holds a row shard, not the whole model")
        mine = dict(sd, V_row=self.V_row)
        for n in self._sharded_names():
            for key in (n, "slot1_" + n, "slot2_" + n):
                if key in sd:
                    mine[key] = sd[key][rank::world]
        self.load_state_dict(mine)

    @property
    def global_bias(self) -> float:
        return float(self.scalars[0].item())

    @property
    def global_step(self) -> int:
        return int(self.step.item())


class TablesView:
    """A glove_tables struct over buffers picked by the caller: the tables of `base` with some of them replaced
    (a fetched block as the col table, a col shard standing on the row side, ...).  Offers what the wrappers use:
    struct(), d, V, V_row, device, optimizer."""

    def __init__(self, base: "DeviceTables", V: int, V_row: int, keep=(), **replace):
        self.d, self.d_model, self.device, self.optimizer = base.d, base.d_model, base.device, base.optimizer
        self.V, self.V_row = int(V), int(V_row)
        self._keep = (base, keep, replace)                      # the struct holds raw pointers: keep the owners alive
        src = base.struct()
        s = GloveTables()
        for name, _ in GloveTables._fields_:
            setattr(s, name, getattr(src, name))
        s.V, s.V_row = self.V, (0 if self.V_row == self.V else self.V_row)
        s.R_ver = s.R_tag = s.C_tag = None            # views are plain: base.struct() above made the tables canonical
        for name, tensor in replace.items():
            _require(tensor, torch.float32)
            setattr(s, name, tensor.data_ptr())
        self._struct = s

    def struct(self) -> GloveTables:
        return self._struct


class Plan:
    """Device-resident dedup index of one batch (see glove_plan in include/glove_hip.h)."""

    INT_FIELDS = ("r_partner", "r_to_c", "r_chunk_id", "r_chunk_start", "r_uniq_slot", "r_uniq_rec",
                  "c_partner", "c_perm", "c_chunk_id", "c_chunk_start", "c_uniq_slot", "c_uniq_rec", "heavy")

    def __init__(self, B: int, V: int, chunk_cap: int, device, cap_chunks: int | None = None,
                 cap_uniq: int | None = None, V_row: int = 0, records: bool | None = None, links: bool = True,
                 own_pairs: bool = True, run_words: bool = False, _no_pairs_ok: bool = False):
        """`own_pairs=False` (a plan with chunk records that glove_plan_build_sorted refills from a dealt epoch): no pair
        arrays of its own — the records carry partner / w / y, the step functions read nothing else."""
        self.B, self.V, self.chunk_cap, self.V_row = int(B), int(V), int(chunk_cap), int(V_row or 0)
        self.cap_chunks = int(B if cap_chunks is None else cap_chunks)
        self.cap_uniq = int(min(B, V) if cap_uniq is None else cap_uniq)
        dev = torch.device(device)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        n = max(self.B, 1)
        self.heavy_chunks = HEAVY_CHUNKS
        self.cap_heavy = 2 * self.B // (self.heavy_chunks * self.chunk_cap) + 2
        self.counts = torch.zeros(8, **i32)
        self.host_counts = [-1] * 8      # unknown until the build has been synchronised
        self.heavy = torch.zeros(self.cap_heavy, **i32)
        # per-chunk records: carried by small (per-step) plans right away, added to big ones when compacted
        # (`records=True`: a staging plan of a big batch on big tables, refilled every step and stepped in a fused form)
        self.r_crec = self.c_crec = None
        if self.B > 0 and (self.B <= RECORDS_AT_BUILD_MAX if records is None else records):
            # (uninitialised: a build writes every record slot a step reads — the poisoned-plan tests — and a capacity-sized
            # record array is hundreds of MB per side at B = 1 M: zero-filling it cost more than the build)
            self.r_crec, self.c_crec = (torch.empty(self.cap_chunks * self.rec_dwords, **i32) for _ in range(2))
        if not own_pairs and self.r_crec is None and not _no_pairs_ok:
            raise ValueError("a plan without pair arrays of its own needs chunk records")
        self.r_partner, self.c_partner = (torch.empty(n, **i32), torch.empty(n, **i32)) if own_pairs else (None, None)
        # c_perm / r_to_c link the two sorted orders; no kernel reads them: `links=False` (the per-step plans of a
        # reshuffled epoch) leaves them out and the build skips the join of its two sorts
        self.r_to_c, self.c_perm = (torch.empty(n, **i32), torch.empty(n, **i32)) if links else (None, None)
        self.r_w, self.r_y = (torch.empty(n, **f32), torch.empty(n, **f32)) if own_pairs else (None, None)
        self.c_w, self.c_y = (torch.empty(n, **f32), torch.empty(n, **f32)) if own_pairs else (None, None)
        self.r_chunk_id, self.c_chunk_id = (torch.empty(max(self.cap_chunks, 1), **i32) for _ in range(2))
        self.r_chunk_start, self.c_chunk_start = (torch.zeros(self.cap_chunks + 1, **i32) for _ in range(2))
        self.r_uniq_slot, self.c_uniq_slot = (torch.zeros(self.cap_uniq + 1, **i32) for _ in range(2))
        self.r_uniq_rec, self.c_uniq_rec = (torch.zeros(4 * max(self.cap_uniq, 1), **i32) for _ in range(2))
        # bitmaps of the batch's ids (glove_plan.r_mark / c_mark): small batches carry them — the one-launch Adam step's sweep
        # over all rows leaves the batch's rows alone by them
        # per-chunk run words (glove_plan.r_chunk_hw): the fused step forms on a plan that keeps pair arrays instead of records
        self.r_chunk_hw = self.c_chunk_hw = None
        if run_words:
            self.r_chunk_hw, self.c_chunk_hw = (torch.zeros(max(self.cap_chunks, 1), **i32) for _ in range(2))
        self.r_mark = self.c_mark = None
        if 0 < self.B <= TAGGED_STEP_MAX_BATCH:
            self.r_mark = torch.zeros(((max(self.V_row, 0) or self.V) + 31) // 32, **i32)
            self.c_mark = torch.zeros((self.V + 31) // 32, **i32)
        self._struct = None

    @property
    def fusable(self) -> bool:
        """The plan carries what the fused step forms read the id layout from: chunk records, or run words beside pair arrays."""
        return self.r_crec is not None or (getattr(self, "r_chunk_hw", None) is not None and
                                           (self.r_partner is not None or getattr(self, "borrows", False)))

    @property
    def rec_dwords(self) -> int:
        """int32 per chunk record in memory (glove_common.h rec_stride_q): whole 128-byte lines — header + block 0 of 8 pairs
        + 16 bytes of padding fill the first, the other blocks follow packed."""
        capP = (self.chunk_cap + 7) // 8 * 8
        return (8 + 6 * (capP // 8 - 1) + 7) // 8 * 8 * 4

    def records(self, side: str, n_chunks: int) -> torch.Tensor:
        """The first n_chunks records of a side ('r' / 'c') in their packed form [n_chunks, 4 + 3 capP]: header {id, pairs, position
        of the id, first-chunk flag | chunks behind}, then capP / 8 blocks of {partner[8] | w[8] | y[8]} (tests and tools)."""
        capP = (self.chunk_cap + 7) // 8 * 8
        raw = getattr(self, side + "_crec")[:n_chunks * self.rec_dwords].view(n_chunks, self.rec_dwords)
        return torch.cat([raw[:, :28], raw[:, 32:32 + 24 * (capP // 8 - 1)]], dim=1)

    def struct(self) -> GlovePlan:
        if self._struct is None:
            s = GlovePlan()
            s.r_crec, s.c_crec = _ptr(self.r_crec), _ptr(self.c_crec)
            s.r_mark, s.c_mark = _ptr(getattr(self, "r_mark", None)), _ptr(getattr(self, "c_mark", None))
            s.r_chunk_hw, s.c_chunk_hw = _ptr(getattr(self, "r_chunk_hw", None)), _ptr(getattr(self, "c_chunk_hw", None))
            s.B, s.chunk_cap, s.cap_chunks, s.cap_uniq = self.B, self.chunk_cap, self.cap_chunks, self.cap_uniq
            s.heavy_chunks, s.cap_heavy, s.V_row = self.heavy_chunks, self.cap_heavy, getattr(self, "V_row", 0)
            s.counts = _ptr(self.counts)
            for i in range(8):
                s.host_counts[i] = self.host_counts[i]
            s.r_w, s.r_y = _ptr(self.r_w), _ptr(self.r_y)
            s.c_w, s.c_y = _ptr(self.c_w), _ptr(self.c_y)
            for n in self.INT_FIELDS:
                setattr(s, n, _ptr(getattr(self, n)))
            self._struct = s
        return self._struct

    def compact(self, lib=None, d: int | None = None, records: bool | None = None) -> "Plan":
        """Exact-size copy (one host sync): used when plans of a static stream stay resident.  With `lib`
        (the loaded C library) the copy also gets its per-chunk records; `d` (floats per table row) tells whether
        the batch is one the library steps in its fused form, which reads the id layout from the records.
        `records`: True / False overrides the rule below (tests and A/B tools)."""
        nc_r, nu_r, nc_c, nu_c, n_heavy, n_mapped = (int(x) for x in self.counts.tolist()[:6])
        if n_mapped:
            logger.warning("%d ids outside [0, %d) were treated as id 0 (the unknown token)", n_mapped, self.V)
        out = Plan.__new__(Plan)
        out.B, out.V, out.chunk_cap, out.V_row = self.B, self.V, self.chunk_cap, self.V_row
        out.cap_chunks, out.cap_uniq = max(nc_r, nc_c), max(nu_r, nu_c)
        out.counts = self.counts.clone()
        most = 0                 # the most chunks of one id (uniq_rec = {id, first chunk, chunks, pairs}); same sync as above
        if self.B > 0:
            most = int(max(self.r_uniq_rec[2:4 * nu_r:4].max().item() if nu_r else 0,
                           self.c_uniq_rec[2:4 * nu_c:4].max().item() if nu_c else 0))
        out.host_counts = [nc_r, nu_r, nc_c, nu_c, n_heavy, -1, most, -1]
        out.heavy_chunks, out.cap_heavy = self.heavy_chunks, max(n_heavy, 1)
        out.heavy = self.heavy[:max(n_heavy, 1)].clone()
        out.r_partner, out.r_w, out.r_y, out.r_to_c = self.r_partner, self.r_w, self.r_y, self.r_to_c
        out.c_partner, out.c_perm, out.c_w, out.c_y = self.c_partner, self.c_perm, self.c_w, self.c_y
        out.r_chunk_id = self.r_chunk_id[:max(out.cap_chunks, 1)].clone()
        out.c_chunk_id = self.c_chunk_id[:max(out.cap_chunks, 1)].clone()
        out.r_chunk_start = self.r_chunk_start[:out.cap_chunks + 1].clone()
        out.c_chunk_start = self.c_chunk_start[:out.cap_chunks + 1].clone()
        out.r_uniq_slot = self.r_uniq_slot[:out.cap_uniq + 1].clone()
        out.c_uniq_slot = self.c_uniq_slot[:out.cap_uniq + 1].clone()
        out.r_uniq_rec = self.r_uniq_rec[:4 * max(out.cap_uniq, 1)].clone()
        out.c_uniq_rec = self.c_uniq_rec[:4 * max(out.cap_uniq, 1)].clone()
        out._struct = None
        out.r_mark, out.c_mark = self.r_mark, self.c_mark
        out.r_chunk_hw = None if self.r_chunk_hw is None else self.r_chunk_hw[:max(out.cap_chunks, 1)].clone()
        out.c_chunk_hw = None if self.c_chunk_hw is None else self.c_chunk_hw[:max(out.cap_chunks, 1)].clone()
        out.r_crec = out.c_crec = None
        # records pad every chunk to the cap: worth it for the latency they save unless the chunks are nearly
        # empty (V = 400 k, B = 1 M: 2.6 pairs per 16-slot chunk -> 7 % more traffic, measured slower)
        fused = d is not None and (nu_r + nu_c) * d * 16 >= FUSED_STEP_BYTES
        if records is None:
            # (small batches always: the tagged step of the latency-bound regime reads nothing but the records)
            records = out.B <= TAGGED_STEP_MAX_BATCH or 4 * out.B >= out.chunk_cap * max(nc_r, nc_c)
            # a fused step reads the id layout from the records or from the run words; with the words it takes the pair fields
            # from the plan's arrays, a group's chunks at a time (V = 400 k, d = 300: 579 against 586 us per step; V = 2 M,
            # d = 128: 509 against 525), and the plan costs 40 instead of 190 B per nonzero
            # (that pays when a lane group owns many chunks — 12 at these sizes — so that the two trips up front are shared; at
            # V = 50 k, d = 300, B = 131,072 a group owns 2 and the records win, 98 against 104 us)
            if fused:
                records = out.r_chunk_hw is None or max(nc_r, nc_c) < RUN_WORDS_MIN_CHUNKS
        if not fused or records:
            out.r_chunk_hw = out.c_chunk_hw = None
        if lib is not None and out.B > 0 and records:
            n = max(out.cap_chunks, 1) * out.rec_dwords
            out.r_crec = torch.empty(n, dtype=torch.int32, device=self.counts.device)
            out.c_crec = torch.empty(n, dtype=torch.int32, device=self.counts.device)
            _check(lib.glove_plan_fill_records(C.byref(out.struct()), _stream()), "glove_plan_fill_records")
        return out

    def nbytes(self) -> int:
        n = sum(getattr(self, f).numel() * 4 for f in self.INT_FIELDS + ("r_w", "r_y", "c_w", "c_y", "counts", "r_mark", "c_mark", "r_chunk_hw", "c_chunk_hw") if getattr(self, f, None) is not None)
        return n + sum(t.numel() * 4 for t in (self.r_crec, self.c_crec) if t is not None)


class Pairs:
    """One sorted order of a set of nonzeros as four device arrays (glove_pairs): the pair's id on the order's own side,
    its id on the other side, glove_weight, glove_value."""

    def __init__(self, n: int, device):
        dev = torch.device(device)
        self.n = int(n)
        self.id = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        self.partner = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        self.w = torch.empty(max(self.n, 1), dtype=torch.float32, device=dev)
        self.y = torch.empty(max(self.n, 1), dtype=torch.float32, device=dev)
        self._struct = None

    def struct(self) -> GlovePairs:
        if self._struct is None:
            s = GlovePairs()
            s.id, s.partner, s.w, s.y = _ptr(self.id), _ptr(self.partner), _ptr(self.w), _ptr(self.y)
            self._struct = s
        return self._struct

    def arrays(self, lo: int = 0, hi: int | None = None):
        hi = self.n if hi is None else hi
        return self.id[lo:hi], self.partner[lo:hi], self.w[lo:hi], self.y[lo:hi]


class Masters:
    """A rank's nonzeros in their two master orders (glove_masters_build): row-major, col-major and the link between them."""

    def __init__(self, row_major: Pairs, col_major: Pairs, link: torch.Tensor, mapped: int):
        self.row_major, self.col_major, self.link, self.mapped = row_major, col_major, link, mapped
        self.n = row_major.n


class PlanBlock:
    """Staging plans that glove_plan_build_sorted refills together: the plans, a host array of their structs and the same
    array in device memory (the kernels read the structs from there: one launch covers the whole block)."""

    def __init__(self, plans: list):
        self.plans = list(plans)
        n = len(self.plans)
        # the plans' counts side by side in ONE tensor: a whole block's counts come back to the host in one copy
        self.counts = torch.zeros(n, 8, dtype=torch.int32, device=self.plans[0].counts.device)
        self.counts_host = torch.zeros(n, 8, dtype=torch.int32).pin_memory() if self.counts.is_cuda else torch.zeros(n, 8, dtype=torch.int32)
        for i, p in enumerate(self.plans):
            p.counts = self.counts[i]
            p._struct = None
        self.host = (GlovePlan * n)()
        for i, p in enumerate(self.plans):
            self.host[i] = p.struct()
        raw = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8)
        self.dev = raw.to(self.plans[0].counts.device)

    def __len__(self):
        return len(self.plans)

    def fetch_counts(self):
        """Starts the copy of every plan's counts to pinned host memory (on the current stream: behind the build that wrote them)."""
        self.counts_host.copy_(self.counts, non_blocking=True)

    def adopt_counts(self, n: int):
        """The fetched counts of the first n plans become their host counts: the step then sizes its grids by what the batch
        holds, like a resident plan's (call once the copy has completed).  The structs a BUILD reads (`host`) keep -1."""
        got = self.counts_host[:n].tolist()
        for p, c in zip(self.plans[:n], got):
            p.host_counts = [c[0], c[1], c[2], c[3], c[4], -1, -1, -1]
            st = p.struct()
            for i in range(8):
                st.host_counts[i] = p.host_counts[i]


def make_hyper(l2_reg=0.01, reg_mult=2.0, learning_rate=0.001, epsilon=1e-7, beta1=0.9, beta2=0.999,
               batch_size=None, inv_batch=None, sides=0, head=HEAD_REGRESSION, neg_factor=1.0,
               step_form=STEP_AUTO, optimizer="Adagrad", momentum=0.0, nesterov=False, rho=None) -> GloveHyper:
    """`sides`: 0/3 both sides, 1 row side only, 2 col side only; `head`: HEAD_REGRESSION (GloVe) or
    HEAD_LOGISTIC (pos/neg logistic matrix factorisation, with `neg_factor`) — see glove_hyper in the header."""
    h = GloveHyper()
    h.sides, h.head, h.neg_factor, h.step_form = sides, head, neg_factor, step_form
    h.optimizer = OPTIMIZER_CODES[optimizer] if isinstance(optimizer, str) else int(optimizer)     # read by glove_step_sparse_f32 only
    if rho is None:                     # the optimizer's own Keras default
        rho = 0.95 if h.optimizer == OPTIMIZER_CODES["Adadelta"] else 0.9
    h.momentum, h.nesterov, h.rho = momentum, int(bool(nesterov)), rho
    h.beta1, h.beta2 = beta1, beta2
    h.l2_reg, h.reg_mult, h.learning_rate, h.epsilon = l2_reg, reg_mult, learning_rate, epsilon
    h.inv_batch = inv_batch if inv_batch is not None else 1.0 / batch_size
    return h


def staging_records(B: int, V_row: int, V: int, d: int) -> bool | None:
    """Whether a staging plan (refilled on the device every step: a reshuffled epoch) should carry chunk records from its
    build: yes where the library will take a fused step form for it (glove_step.hip pick_step_form judges such a plan by
    the most ids its batch can hold), otherwise the Plan's own rule (small batches)."""
    return True if min(V_row + V, 2 * B) * ((d + 3) // 4 * 4) * 16 >= FUSED_STEP_BYTES else None


def _step_struct(tables, plans, hyper):
    """The tables as the Adagrad step may see them: a twinned row table stays twinned between steps.  Marks the tables
    as possibly twinned when one of the plans can take the twin form (the rule of glove_step.hip pick_step_form), so
    that the next reader of R / br brings them home first."""
    if getattr(tables, "R_tag", None) is not None:
        # step-tagged tables: the tagged step leaves rows in their second copies (the rule of glove_step.hip pick_step_form;
        # Adam: adam_one_launch)
        form = hyper.step_form
        if form == STEP_TAGGED or (form == STEP_AUTO and any(p.r_crec is not None and p.B <= TAGGED_STEP_MAX_BATCH for p in plans)):
            tables._twin_dirty = True
        return tables.struct(twin_ok=True)
    if getattr(tables, "R_ver", None) is None:
        return tables.struct()
    form = hyper.step_form
    for plan in plans:
        hc = plan.host_counts
        ids = hc[1] + hc[3] if hc[1] >= 0 and hc[3] >= 0 else min(tables.V_row + tables.V, 2 * plan.B)
        fused = plan.fusable and ids * tables.d * 16 >= FUSED_STEP_BYTES
        if form == STEP_FUSED_TWIN or (form == STEP_AUTO and fused):
            tables._twin_dirty = True
            break
    return tables.struct(twin_ok=True)


class GloveHip:
    """Thin object wrapper over the C ABI; one instance per process/GPU."""

    def __init__(self, device="cuda:0", lib_path=None, any_abi=False):
        self.lib = load_library(lib_path, any_abi)
        self.device = torch.device(device)
        self._plan_ws = None
        self._step_ws = None
        self.ws_generation = 0      # counts reallocations of the shared workspaces: a captured hipGraph holds their raw pointers

    # ---- workspaces (grown on demand, never inside a captured region)
    def _ws(self, attr: str, nbytes: int) -> torch.Tensor:
        cur = getattr(self, attr)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            setattr(self, attr, cur)
            self.ws_generation += 1
        return cur

    def step_workspace(self, plan: Plan, d: int) -> torch.Tensor:
        return self._ws("_step_ws", self.lib.glove_step_workspace_bytes(plan.B, plan.cap_chunks, d))

    # ---- index build
    def build_plan(self, row, col, w, y, V: int, chunk_cap: int | None = DEFAULT_CHUNK_CAP, compact=False,
                   into: Plan | None = None, ws: torch.Tensor | None = None, d: int | None = None,
                   V_row: int = 0, records: bool | None = None, links: bool = True, run_words: bool | None = None) -> Plan:
        """Builds the dedup index of one batch on the device.  `V_row`: rows of this rank's row-table shard when the
        row ids are shard-local (ids outside it count as id 0, like col ids outside [0, V)).  `run_words`: the per-chunk run words a
        fused step needs of a plan without records (default: when `d` says the batch can reach the fused regime).  `into`: a full-capacity Plan of the same
        (B, V, chunk_cap) to refill — a caller that indexes a fresh batch every step avoids ~20 tensor
        allocations per step this way.  `ws`: scratch of glove_plan_workspace_bytes(B, V) bytes (default: one
        shared buffer, fine for builds issued on one stream).  `records`: see Plan (default: small batches only)."""
        B = int(row.numel())
        _require(row, torch.int32, B); _require(col, torch.int32, B)
        _require(w, torch.float32, B); _require(y, torch.float32, B)
        if not chunk_cap:
            chunk_cap = auto_chunk_cap(B, V, d)
        if into is not None:
            if (into.B, into.V, into.chunk_cap) != (B, V, chunk_cap) or into.cap_chunks < B:
                raise ValueError("`into` must be an uncompacted plan of the same batch size, vocabulary and chunk cap")
            plan = into
        else:
            if run_words is None:
                run_words = d is not None and min((V_row or V) + V, 2 * B) * ((d + 3) // 4 * 4) * 16 >= FUSED_STEP_BYTES
            plan = Plan(B, V, chunk_cap, row.device, V_row=V_row, records=records, links=links, run_words=bool(run_words))
        if ws is None:      # builds that run concurrently on different streams each bring their own scratch
            ws = self._ws("_plan_ws", self.lib.glove_plan_workspace_bytes(B, V))
        _check(self.lib.glove_plan_build(_ptr(row), _ptr(col), _ptr(w), _ptr(y), B, V, C.byref(plan.struct()),
                                         _ptr(ws), ws.numel(), _stream()), "glove_plan_build")
        return plan.compact(self.lib, d) if compact else plan

    # ---- epochs dealt from id-sorted master orders (include/glove_hip.h)
    def build_masters(self, row, col, w, y, V: int, V_row: int = 0) -> Masters:
        """The rank's nonzeros sorted once: row-major and col-major orders + the link between them (one host sync for the
        count of ids that were mapped to 0)."""
        n = int(row.numel())
        _require(row, torch.int32, n); _require(col, torch.int32, n)
        _require(w, torch.float32, n); _require(y, torch.float32, n)
        dev = row.device
        rm, cm = Pairs(n, dev), Pairs(n, dev)
        link = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        mapped = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(self.lib.glove_masters_workspace_bytes(n), 256), dtype=torch.uint8, device=dev)
        _check(self.lib.glove_masters_build(_ptr(row), _ptr(col), _ptr(w), _ptr(y), n, V, V_row, C.byref(rm.struct()),
                                            C.byref(cm.struct()), _ptr(link), _ptr(mapped), _ptr(ws), ws.numel(), _stream()),
               "glove_masters_build")
        n_mapped = int(mapped.item())
        if n_mapped:
            logger.warning("%d ids outside their table were treated as id 0 (the unknown token)", n_mapped)
        del ws
        return Masters(rm, cm, link, n_mapped)

    def deal_workspace(self, n: int, B: int, device) -> torch.Tensor:
        return torch.empty(max(self.lib.glove_epoch_deal_workspace_bytes(n, B), 256), dtype=torch.uint8, device=device)

    def deal_epoch(self, masters: Masters, B: int, key: int, row_side: Pairs, col_side: Pairs, ws: torch.Tensor) -> None:
        """One epoch of `masters` under the bijection the 128-bit `key` determines, into row_side / col_side: batch k =
        positions [k B, (k + 1) B) of both, sorted by row id / by col id."""
        if row_side.n != masters.n or col_side.n != masters.n:
            raise GloveHipError("epoch buffers must hold %d pairs" % masters.n)
        _check(self.lib.glove_epoch_deal(C.byref(masters.row_major.struct()), C.byref(masters.col_major.struct()),
                                         _ptr(masters.link), masters.n, B, key & (2 ** 64 - 1), (key >> 64) & (2 ** 64 - 1),
                                         C.byref(row_side.struct()), C.byref(col_side.struct()), _ptr(ws), ws.numel(), _stream()),
               "glove_epoch_deal")

    def staging_plan(self, B: int, V: int, chunk_cap: int, device, V_row: int = 0, records: bool = True, run_words: bool = False,
                     borrow: bool = False) -> Plan:
        """A plan for glove_plan_build_sorted to refill: capacity for any batch of B pairs (an id of p pairs has at most
        p / chunk_cap + 1 chunks), chunk records carrying the pair fields, or pair arrays of its own without records
        (`run_words`: with the per-chunk run words the fused step forms need in that case; `borrow`: no pair arrays either —
        build_plans_sorted points the plan at the batch's positions in the epoch's arrays, which must outlive its steps)."""
        cap_uniq = min(B, max(V, V_row or 0))
        cap_chunks = int(self.lib.glove_plan_chunk_bound(B, cap_uniq, chunk_cap))
        words = run_words and not records
        plan = Plan(B, V, chunk_cap, device, cap_chunks=cap_chunks, cap_uniq=cap_uniq, V_row=V_row, records=bool(records),
                    links=False, own_pairs=not records and not (borrow and words), run_words=words,
                    _no_pairs_ok=bool(borrow and words))
        plan.borrows = bool(borrow and words)
        return plan

    def build_plans_sorted(self, row_side: Pairs, col_side: Pairs, first_batch: int, block: PlanBlock, n: int,
                           V: int, ws: torch.Tensor) -> None:
        """The indexes of batches first_batch .. first_batch + n - 1 of a dealt epoch into the first n plans of `block`."""
        B = block.plans[0].B
        if n < 0 or n > len(block) or (first_batch + n) * B > row_side.n:
            raise GloveHipError("batches %d .. %d of %d pairs do not lie inside the epoch" % (first_batch, first_batch + n - 1, B))
        _check(self.lib.glove_plan_build_sorted(C.byref(row_side.struct()), C.byref(col_side.struct()), first_batch * B, B, n, V,
                                                block.host, _ptr(block.dev), _ptr(ws), ws.numel(), _stream()),
               "glove_plan_build_sorted")
        if getattr(block.plans[0], "borrows", False):
            # plans that borrow their pair fields: batch j lies sorted at positions [(first_batch + j) B, ...) of the epoch's arrays
            base = [(t.data_ptr(), t) for t in (row_side.partner, row_side.w, row_side.y, col_side.partner, col_side.w, col_side.y)]
            for j in range(n):
                st, off = block.plans[j].struct(), 4 * (first_batch + j) * B
                st.r_partner, st.r_w, st.r_y = (base[k][0] + off for k in range(3))
                st.c_partner, st.c_w, st.c_y = (base[k][0] + off for k in range(3, 6))
                block.plans[j].lent = (row_side, col_side)          # (keeps the epoch's arrays alive as long as the plan points at them)

    # ---- passes
    def passes(self, plan, tables, hyper, ws=None):
        """Both gather passes (row side and col side) in one launch."""
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_passes_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper),
                                         _ptr(ws), ws.numel(), _stream()), "glove_passes_f32")

    def rowpass(self, plan, tables, hyper, ws=None):
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_rowpass_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper),
                                          _ptr(ws), ws.numel(), _stream()), "glove_rowpass_f32")

    def colpass(self, plan, tables, hyper, ws=None):
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_colpass_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper),
                                          _ptr(ws), ws.numel(), _stream()), "glove_colpass_f32")

    def rowside_step(self, plan, tables, hyper, ws=None):
        """The row side of a step, applied in place where a lane group holds an id completely (hyper.sides = 1); the
        col pass of the step must have run already: it gathers the old rows."""
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_rowside_step_adagrad_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper),
                                                       _ptr(ws), ws.numel(), _stream()), "glove_rowside_step_adagrad_f32")

    def apply_adagrad(self, plan, tables, hyper, loss_out=None, ws=None):
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_apply_adagrad_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper),
                                                _ptr(ws), ws.numel(), _ptr(loss_out), _stream()),
               "glove_apply_adagrad_f32")

    def dense_grad(self, plan, tables, hyper, G_flat, ws=None):
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_dense_grad_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper),
                                             _ptr(ws), ws.numel(), _ptr(G_flat), _stream()), "glove_dense_grad_f32")

    def dense_adagrad(self, tables, hyper, G_flat, loss_out=None):
        _check(self.lib.glove_dense_adagrad_f32(C.byref(tables.struct()), C.byref(hyper), _ptr(G_flat),
                                                _ptr(loss_out), _stream()), "glove_dense_adagrad_f32")

    def dense_adam(self, tables, hyper, G_flat, loss_out=None):
        _check(self.lib.glove_dense_adam_f32(C.byref(tables.struct()), C.byref(hyper), _ptr(G_flat),
                                             _ptr(loss_out), _stream()), "glove_dense_adam_f32")

    def grad_layout(self, tables) -> dict:
        """Float offsets of the sections of the flat dense-gradient buffer."""
        offs = (C.c_int64 * 5)()
        total = self.lib.glove_dense_grad_layout(tables.V_row, tables.V, tables.d, C.cast(offs, C.c_void_p))
        return dict(G_R=offs[0], G_br=offs[1], G_C=offs[2], G_bc=offs[3], tail=offs[4], total=int(total))

    def dense_grad_buffer(self, tables) -> torch.Tensor:
        return torch.zeros(self.grad_layout(tables)["total"], dtype=torch.float32, device=tables.device)

    # ---- touched-rows exchange (include/glove_hip.h "touched-rows exchange")
    def packed_list(self, buf: torch.Tensor, with_header=True, ids: torch.Tensor | None = None, n: int = -1,
                    side: int = -1) -> GlovePackedList:
        """One list inside `buf` ([entries, d + 4] float32, contiguous): header in entry 0 and the count read from it on
        the device (with_header), or a bare run of `n` entries, optionally with explicit owner-local `ids`."""
        _require(buf, torch.float32)
        stride = buf.shape[-1]
        lst = GlovePackedList()
        lst.entries = buf.data_ptr() + (4 * stride if with_header else 0)
        lst.header = buf.data_ptr() if with_header else None
        if ids is not None:
            _require(ids, torch.int32)
            if ids.numel() < n:
                raise GloveHipError("%d ids for %d entries" % (ids.numel(), n))
        lst.ids = _ptr(ids)
        lst.n, lst.side = n, side
        return lst

    def pack_grad(self, plan, tables, hyper, packed: torch.Tensor, ws=None):
        """The plan's summed gradients (hyper.sides) as one packed list: packed is [capacity, d + 4] float32."""
        _require(packed, torch.float32)
        if packed.dim() != 2 or packed.shape[1] != tables.d + 4:
            raise GloveHipError("packed buffer must be [entries, d + 4]")
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_pack_grad_f32(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper), _ptr(ws),
                                            ws.numel(), _ptr(packed), packed.shape[0], _stream()), "glove_pack_grad_f32")

    def _packing_call(self, name, plan, tables, hyper, packed, ws):
        _require(packed, torch.float32)
        if packed.dim() != 2 or packed.shape[1] != tables.d + 4:
            raise GloveHipError("packed buffer must be [entries, d + 4]")
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(getattr(self.lib, name)(C.byref(plan.struct()), C.byref(tables.struct()), C.byref(hyper), _ptr(ws),
                                       ws.numel(), _ptr(packed), packed.shape[0], _stream()), name)

    def passes_packing(self, plan, tables, hyper, packed: torch.Tensor, ws=None):
        """The passes of hyper.sides; ids one lane group holds completely land in the packed list right away."""
        self._packing_call("glove_passes_packing_f32", plan, tables, hyper, packed, ws)

    def pack_rest(self, plan, tables, hyper, packed: torch.Tensor, ws=None):
        """Completes the list passes_packing started (the other ids, the header)."""
        self._packing_call("glove_pack_rest_f32", plan, tables, hyper, packed, ws)

    def loss_partials(self, plan, tables, out4: torch.Tensor, ws=None):
        """out4 = {sum e, sum w diff^2, sum |r|^2+|c|^2, sum b^2} of the plan's last row pass (a list header's floats 2..5)."""
        _require(out4, torch.float32, 4)
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_loss_partials_f32(C.byref(plan.struct()), C.byref(tables.struct()), _ptr(ws), ws.numel(),
                                                _ptr(out4), _stream()), "glove_loss_partials_f32")

    def combine_packed(self, lst: GlovePackedList, tag: int, tables, G_flat, mark, capacity: int):
        _require(G_flat, torch.float32)
        _require(mark, torch.int32)
        if mark.numel() < tables.V_row + tables.V:
            raise GloveHipError("mark needs V_row + V entries")
        _check(self.lib.glove_combine_packed_f32(C.byref(lst), tag, C.byref(tables.struct()), _ptr(G_flat), _ptr(mark),
                                                 capacity, _stream()), "glove_combine_packed_f32")

    def count_packed(self, lists, tables, G_flat, mark, capacity: int = 0):
        """Optional before the combines: ids only one list touches are then applied straight from their entry."""
        arr = (GlovePackedList * len(lists))(*lists)
        _check(self.lib.glove_count_packed_f32(arr, len(lists), C.byref(tables.struct()), _ptr(G_flat), _ptr(mark), capacity,
                                               _stream()), "glove_count_packed_f32")

    def apply_packed(self, lists, tables, hyper, G_flat, mark, tail=None, loss_out=None, capacity: int = 0):
        arr = (GlovePackedList * len(lists))(*lists)
        _check(self.lib.glove_apply_packed_adagrad_f32(arr, len(lists), C.byref(tables.struct()), C.byref(hyper),
                                                       _ptr(G_flat), _ptr(mark), _ptr(tail), _ptr(loss_out), capacity,
                                                       _stream()), "glove_apply_packed_adagrad_f32")

    def gather_rows(self, W, bias, ids, rows, biases):
        for x in (W, bias, rows, biases):
            _require(x, torch.float32)
        _require(ids, torch.int32)
        n = int(ids.numel())
        if rows.numel() < n * W.shape[1] or biases.numel() < n:
            raise GloveHipError("gather_rows: output buffers too small")
        _check(self.lib.glove_gather_rows_f32(_ptr(W), _ptr(bias), _ptr(ids), n, W.shape[1], _ptr(rows), _ptr(biases),
                                              _stream()), "glove_gather_rows_f32")

    # ---- whole steps
    def step_adagrad(self, plan, tables, hyper, loss_out=None, ws=None):
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_step_adagrad_f32(C.byref(plan.struct()), C.byref(_step_struct(tables, (plan,), hyper)), C.byref(hyper),
                                               _ptr(ws), ws.numel(), _ptr(loss_out), _stream()),
               "glove_step_adagrad_f32")

    def steps_adagrad(self, plans, tables, hyper, loss_out=None, ws=None):
        """len(plans) consecutive Adagrad steps from one host call (the launch loop runs in C; on step-tagged tables consecutive
        small batches go out as a chain: ONE launch per step, glove_steps_adagrad_f32)."""
        if not plans:
            return
        if ws is None:
            big = max(plans, key=lambda p: self.lib.glove_step_workspace_bytes(p.B, p.cap_chunks, tables.d))
            ws = self.step_workspace(big, tables.d)
        arr = (C.POINTER(GlovePlan) * len(plans))(*[C.pointer(p.struct()) for p in plans])
        _check(self.lib.glove_steps_adagrad_f32(arr, len(plans), C.byref(_step_struct(tables, plans, hyper)), C.byref(hyper), _ptr(ws),
                                                ws.numel(), _ptr(loss_out), _stream()), "glove_steps_adagrad_f32")

    def step_adam(self, plan, tables, hyper, G_flat, loss_out=None, ws=None):
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        _check(self.lib.glove_step_adam_f32(C.byref(plan.struct()), C.byref(_step_struct(tables, (plan,), hyper)), C.byref(hyper),
                                            _ptr(ws), ws.numel(), _ptr(G_flat), _ptr(loss_out), _stream()),
               "glove_step_adam_f32")

    def step_sparse(self, plan, tables, hyper, G_flat=None, loss_out=None, ws=None):
        """One step under the Keras optimizer `tables.optimizer` names (glove_step_sparse_f32: SGD, RMSprop, Adamax, Adadelta, Ftrl, Nadam; Adagrad
        and Adam go to their own entry points).  G_flat: the dense gradient buffer RMSprop, Nadam and Adam need."""
        ws = self.step_workspace(plan, tables.d) if ws is None else ws
        hyper.optimizer = OPTIMIZER_CODES[tables.optimizer]
        struct = _step_struct(tables, (plan,), hyper) if tables.optimizer in ("Adagrad", "Adam") else tables.struct()
        _check(self.lib.glove_step_sparse_f32(C.byref(plan.struct()), C.byref(struct), C.byref(hyper), _ptr(ws), ws.numel(),
                                              _ptr(G_flat), _ptr(loss_out), _stream()), "glove_step_sparse_f32")

    def steps_adam(self, plans, tables, hyper, G_flat, loss_out=None, ws=None):
        """len(plans) consecutive Adam steps from one host call (on twinned tables consecutive small batches go out as a chain:
        ONE launch per step, glove_steps_adam_f32)."""
        if not plans:
            return
        if ws is None:
            big = max(plans, key=lambda p: self.lib.glove_step_workspace_bytes(p.B, p.cap_chunks, tables.d))
            ws = self.step_workspace(big, tables.d)
        arr = (C.POINTER(GlovePlan) * len(plans))(*[C.pointer(p.struct()) for p in plans])
        _check(self.lib.glove_steps_adam_f32(arr, len(plans), C.byref(_step_struct(tables, plans, hyper)), C.byref(hyper), _ptr(ws),
                                             ws.numel(), _ptr(G_flat), _ptr(loss_out), _stream()),
               "glove_steps_adam_f32")

    # ---- eval / predict
    def eval_sums(self, row, col, w, y, tables, sums=None) -> torch.Tensor:
        n = int(row.numel())
        _require(row, torch.int32, n); _require(col, torch.int32, n)
        _require(w, torch.float32, n); _require(y, torch.float32, n)
        _require(sums, torch.float64)
        if sums is None:
            sums = torch.zeros(4, dtype=torch.float64, device=tables.device)
        _check(self.lib.glove_eval_f32(_ptr(row), _ptr(col), _ptr(w), _ptr(y), int(row.numel()),
                                       C.byref(tables.struct()), _ptr(sums), _stream()), "glove_eval_f32")
        return sums

    def eval_sums_logistic(self, row, col, pos, neg, tables, sums=None) -> torch.Tensor:
        n = int(row.numel())
        _require(row, torch.int32, n); _require(col, torch.int32, n)
        _require(pos, torch.float32, n); _require(neg, torch.float32, n)
        _require(sums, torch.float64)
        if sums is None:
            sums = torch.zeros(6, dtype=torch.float64, device=tables.device)
        _check(self.lib.glove_eval_logistic_f32(_ptr(row), _ptr(col), _ptr(pos), _ptr(neg), int(row.numel()),
                                                C.byref(tables.struct()), _ptr(sums), _stream()),
               "glove_eval_logistic_f32")
        return sums

    def topk_cosine(self, R: torch.Tensor, query_ids: torch.Tensor, k: int):
        _require(R, torch.float32); _require(query_ids, torch.int32)
        V, d = R.shape
        n = int(query_ids.numel())
        sims = torch.empty(n, k, dtype=torch.float32, device=R.device)
        idx = torch.empty(n, k, dtype=torch.int32, device=R.device)
        ws = torch.empty(self.lib.glove_topk_workspace_bytes(n, V, k), dtype=torch.uint8, device=R.device)
        _check(self.lib.glove_topk_cosine_f32(_ptr(R), V, d, _ptr(query_ids), n, k, _ptr(sims), _ptr(idx),
                                              _ptr(ws), ws.numel(), _stream()), "glove_topk_cosine_f32")
        return sims, idx

    # ---- data prep
    def cooccurrence(self, tokens: torch.Tensor, V: int, context: int, cap: int | None = None):
        """(row i32, col i32, count i64, value f64) sorted by (row, col): the symmetrised window
        co-occurrence table of reference src/data/text8.py:84-108 (one host sync for the size)."""
        _require(tokens, torch.int32)
        n = int(tokens.numel())
        cap = int(cap if cap is not None else max(1, 2 * context * n))
        dev = tokens.device
        row = torch.empty(cap, dtype=torch.int32, device=dev)
        col = torch.empty(cap, dtype=torch.int32, device=dev)
        cnt = torch.empty(cap, dtype=torch.int64, device=dev)
        val = torch.empty(cap, dtype=torch.float64, device=dev)
        nnz = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = torch.empty(self.lib.glove_cooc_workspace_bytes(n, context), dtype=torch.uint8, device=dev)
        _check(self.lib.glove_cooccurrence_i32(_ptr(tokens), n, V, context, _ptr(row), _ptr(col), _ptr(cnt), _ptr(val),
                                               _ptr(nnz), cap, _ptr(ws), ws.numel(), _stream()),
               "glove_cooccurrence_i32")
        m = int(nnz.item())
        if m > cap:
            raise GloveHipError("co-occurrence table has %d entries, capacity %d" % (m, cap))
        return row[:m], col[:m], cnt[:m], val[:m]
