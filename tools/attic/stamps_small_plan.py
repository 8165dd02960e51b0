#!/usr/bin/env python3
"""Diagnostic: wall-clock stamps of the phases of the one-workgroup index build (libglove_hip_diag.so,
`make -C glove-tensorflow_amd/csrc diag`).  Shares of a diagnostic build only (every stamp drains the wave's loads and stores)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
import torch  # noqa: E402
from helpers import make_batch, to_dev  # noqa: E402
from trainer.hip_api import PKG_DIR, GloveHip, Plan  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
V, cap = 10000, 16
hip = GloveHip("cuda:0", lib_path=PKG_DIR / "lib" / "libglove_hip_diag.so")
hip.lib.glove_debug_set_small_stamps.argtypes = [C.c_void_p]
batches = [to_dev(*make_batch(s, B, V)) for s in range(4)]
staging = Plan(B, V, cap, "cuda:0")
for bt in batches:
    hip.build_plan(*bt, V, chunk_cap=cap, into=staging)
torch.cuda.synchronize()
stamps = torch.zeros(16 * 16, dtype=torch.int64, device="cuda:0")
assert hip.lib.glove_debug_set_small_stamps(stamps.data_ptr()) == 0
# (up to 2,048 pairs waves 0-7 build the row side while waves 8-15 build the col side: the phases are those of either team;
#  beyond, the second side's stamps overwrite the first's)
names = ["start", "batch staged in LDS + ids read back", "sort", "gather partner/w/y + store", "side numbered", "end"]
acc = []
for rep in range(8):
    stamps.zero_()
    torch.cuda.synchronize()
    hip.build_plan(*batches[rep % 4], V, chunk_cap=cap, into=staging)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(16, 16)[:, :6].astype(np.float64)
    st = st[st[:, 0] > 0]                                          # the waves of this workgroup size
    acc.append((st[:, 1:] - st[:, :-1]).max(axis=0) / 100.0)       # 100 MHz -> us; the slowest wave of each phase
acc = np.median(np.array(acc), axis=0)
print("B = %d: phases of plan_small_kernel (us, median of 8 builds, slowest wave; diagnostic build)" % B)
for n, v in zip(names[1:], acc):
    print("  %-28s %6.2f" % (n, v))
print("  %-28s %6.2f" % ("sum", acc.sum()))
