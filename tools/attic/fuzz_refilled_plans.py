#!/usr/bin/env python3
"""Fuzz of the device-refilled staging plan (a reshuffled epoch): random (B, V, d, chunk cap, id distribution, ids outside the
tables), every case built twice into one 0xFF-poisoned full-capacity plan WITH chunk records, compared bit for bit with
oracle/glove_ref.py:build_plan (every record header and slot), range-checked on the device, then stepped — counts never read
back — in the two-launch form and the fused forms (slots / three launches / twin) against the float64 oracle step.
tools/fuzz_refilled_plans.py [cases] [seed]"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "oracle", REPO / "tests"):
    sys.path.insert(0, str(p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import glove_ref as ref  # noqa: E402
from helpers import assert_tables_close, make_batch, oracle_tables, tables_from_oracle, to_dev  # noqa: E402
from helpers import _assert_plan_equals_oracle, _poison  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, Plan, make_hyper  # noqa: E402
import ctypes as C  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
hip = GloveHip("cuda:0")
checks = C.CDLL(str(REPO / "tests" / "native" / "libglove_test_checks.so"))
checks.glove_test_check_plan.restype = C.c_int
errors = torch.zeros(8, dtype=torch.int32, device="cuda:0")
for case in range(cases):
    B = int(rng.choice([rng.integers(1, 64), rng.integers(64, 4097), rng.integers(4097, 12000), rng.integers(12000, 40000)]))
    V = int(rng.choice([rng.integers(2, 40), rng.integers(40, 3000), rng.integers(3000, 70000)]))
    d = int(rng.choice([4, 20, 50, 64, 128, 300]))
    cap = int(rng.choice([1, 2, 3, 8, 16, 32]))
    zipf = bool(rng.integers(0, 2))
    staging = Plan(B, V, cap, "cuda:0", records=True, links=bool(rng.integers(0, 2)))
    words = Plan(B, V, cap, "cuda:0", records=False, run_words=True, links=False)     # pair arrays + run words: the fused forms without records
    ws = torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device="cuda:0")
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    devs = {form: tables_from_oracle(t, DeviceTables) for form in (1, 2, 3, 4, "words 2", "words 3", "words 4")}
    devs[4].enable_twin()
    devs["words 4"].enable_twin()
    for k in range(2):
        row, col, w, y = make_batch(1000 * case + k + seed, B, V, zipf=zipf)
        if rng.integers(0, 3) == 0:
            row[:: int(rng.integers(3, 11))] = V + int(rng.integers(0, 9))
            col[1:: int(rng.integers(3, 11))] = -int(rng.integers(1, 9))
        _poison(staging, ws)
        hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, into=staging, ws=ws)
        rc = checks.glove_test_check_plan(C.byref(staging.struct()), V, C.c_void_p(errors.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        _assert_plan_equals_oracle(staging, ref.build_plan(row, col, cap, V=V), B, w, y)
        assert errors.tolist() == [0] * 8, (case, errors.tolist())
        assert staging.host_counts[1] == -1
        rowc, colc = np.where((row < 0) | (row >= V), 0, row), np.where((col < 0) | (col >= V), 0, col)
        ref.train_step(t, rowc, colc, w, y, hp)
        _poison(words, ws)
        hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, into=words, ws=ws)
        _assert_plan_equals_oracle(words, ref.build_plan(row, col, cap, V=V), B, w, y)
        for form, dt in devs.items():
            plan, f = (words, int(form.split()[1])) if isinstance(form, str) else (staging, form)
            h = make_hyper(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, batch_size=B, step_form=f)
            hip.step_adagrad(plan, dt, h)
            assert_tables_close(dt, t, 2e-5, 2e-6)
        for f in (2, 3, 4):                                     # records or run words: the same bits
            for n in ("R", "C", "br", "bc"):
                assert torch.equal(getattr(devs[f], n), getattr(devs["words %d" % f], n)), (case, f, n)
    if case % 20 == 0:
        print("case %d ok (B=%d V=%d d=%d cap=%d zipf=%s)" % (case, B, V, d, cap, zipf), flush=True)
print("%d cases ok" % cases)
