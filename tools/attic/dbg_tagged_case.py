#!/usr/bin/env python3
"""One tagged Adagrad step against the two-launch form on one batch: which rows differ, and what they have in common
(debugging aid).  Usage: tools/dbg_tagged_case.py B V d cap"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glove-tensorflow_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import glove_ref as ref                                     # noqa: E402  (debugging aid: the oracle is the checker)
from helpers import make_batch, oracle_tables, tables_from_oracle, to_dev        # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper                  # noqa: E402


def main():
    B, V, d, cap = (int(x) for x in sys.argv[1:5])
    hip = GloveHip(torch.device("cuda:0"))
    hp = ref.Hyper(learning_rate=0.05)
    row, col, w, y = make_batch(B + V + d, B, V)
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    a.enable_tags()
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, records=True)
    mk = lambda f: make_hyper(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, epsilon=hp.epsilon, batch_size=B, step_form=f)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    hip.step_adagrad(plan, a, mk(5), la)
    torch.cuda.synchronize()
    raw_R, raw_tag = a._R.clone(), a.R_tag.clone()
    hip.step_adagrad(plan, b, mk(1), lb)
    want = ref.build_plan(row, col, cap, V=V)
    print("loss", la.tolist(), lb.tolist())
    for side, name in (("r", "R"), ("c", "C")):
        A, Bm = getattr(a, name), getattr(b, name)
        bad = ((A - Bm).abs() > 1e-7).any(1).nonzero().flatten().cpu().numpy()
        rec = want[side + "_uniq_rec"]
        ids = rec[:, 0]
        chunks = dict(zip(ids.tolist(), rec[:, 2].tolist()))
        pairs = dict(zip(ids.tolist(), rec[:, 3].tolist()))
        print(name, "rows that differ:", len(bad), "of", len(ids), "ids in the batch; in batch:", int(np.isin(bad, ids).sum()))
        print("  first:", bad[:20].tolist())
        print("  chunks of the bad ids:", sorted(set(chunks.get(int(u), 0) for u in bad)), "pairs:", sorted(set(pairs.get(int(u), 0) for u in bad))[:20])
        if len(bad):
            u = int(bad[0])
            print("  row", u, "tagged", A[u, :4].tolist(), "two-launch", Bm[u, :4].tolist(), "start", torch.from_numpy(np.asarray(getattr(t, name)[u, :4])).tolist())
    print("scalars", a.scalars.tolist())


if __name__ == "__main__":
    main()
