"""Synthetic co-occurrence nonzeros (there is no text8 on disk and no network).

Two generators, both seeded and cheap enough to run inside bench.py on the GPU box:

* `text8_shaped`  — the statistics of an i.i.d. Zipf corpus pushed through the reference's data
  prep (reference src/data/text8.py:84-139): for every unordered token pair the symmetrised
  window count is Poisson(2 * context * n_tokens * p_a * p_b); pairs with count >= 10 are kept,
  each (row, col) appears at most once, and both orientations are present, as in
  `interaction.csv`.  `<UNK>` is id 0 with the 10 % mass coverage 0.9 leaves out of vocabulary.
* `zipf_sampled`  — SURVEY.md §8d's generator for the large configs (V = 50k..2M): row and col
  drawn independently from Zipf(s) over V ranks, row != col, count ~ 10 + floor(Pareto(1.2)).

Both return int32 ids and fp32 `glove_weight = clip((count/100)^0.75, 0, 1)` /
`glove_value = ln(value)` exactly as reference text8.py:129-139 defines them.
"""
from __future__ import annotations

import torch


def glove_weight(count: torch.Tensor, alpha=0.75, x_max=100.0) -> torch.Tensor:
    return (count.double() / x_max).pow(alpha).clamp(0, 1).float()


def _finish(row, col, count, gen, device):
    # value = sum of 1/distance over the pair's occurrences; for a 5-token window the mean of
    # 1/distance is 0.457 (README sample rows: 0.31..0.71)
    u = torch.empty(count.shape, device=count.device).uniform_(0.35, 0.6, generator=gen)
    value = (count.double() * u.double()).clamp_min(1e-3)
    perm = torch.randperm(row.numel(), generator=gen, device=row.device)   # text8.py:118-123 hash-shuffle
    row, col = row[perm].int(), col[perm].int()
    w = glove_weight(count[perm])
    y = value[perm].log().float()
    return row.to(device), col.to(device), w.to(device), y.to(device)


def text8_shaped(V=10000, n_tokens=17_005_207, context=5, zipf_s=1.0, unk_mass=0.1, min_count=10, seed=0,
                 device="cpu", work_device=None):
    """(row, col, weight, value) of a text8-like corpus; about 1e6 nonzeros at the defaults."""
    wd = torch.device(work_device or device)
    gen = torch.Generator(device=wd)
    gen.manual_seed(seed)
    ranks = torch.arange(1, V, dtype=torch.float64, device=wd)
    p = ranks.pow(-zipf_s)
    p = torch.cat([torch.tensor([unk_mass], dtype=torch.float64, device=wd), p / p.sum() * (1.0 - unk_mass)])
    scale = 2.0 * context * n_tokens
    rows, cols, counts = [], [], []
    step = max(1, (1 << 24) // V)
    for a0 in range(0, V, step):
        a = torch.arange(a0, min(a0 + step, V), device=wd)
        lam = scale * p[a][:, None] * p[None, :]
        lam = torch.triu(lam, diagonal=a0 + 1)            # a < b only; symmetrised below
        if float(lam.max()) < 1.0:
            break                                          # rows are sorted by frequency
        c = torch.poisson(lam.float(), generator=gen)
        ia, ib = torch.nonzero(c >= min_count, as_tuple=True)
        rows.append(a[ia]); cols.append(ib); counts.append(c[ia, ib])
    r, c_, n = torch.cat(rows), torch.cat(cols), torch.cat(counts)
    row = torch.cat([r, c_]); col = torch.cat([c_, r]); count = torch.cat([n, n])
    return _finish(row, col, count, gen, device)


def zipf_sampled(V: int, nnz: int, zipf_s=1.0, seed=0, device="cpu", work_device=None):
    wd = torch.device(work_device or device)
    gen = torch.Generator(device=wd)
    gen.manual_seed(seed)
    cdf = torch.arange(1, V + 1, dtype=torch.float64, device=wd).pow(-zipf_s).cumsum(0)
    cdf = cdf / cdf[-1]

    def draw(n):
        u = torch.rand(n, dtype=torch.float64, device=wd, generator=gen)
        return torch.searchsorted(cdf, u).clamp_(max=V - 1)

    row, col = draw(nnz), draw(nnz)
    clash = row == col
    col[clash] = (col[clash] + 1) % V
    u = torch.rand(nnz, dtype=torch.float64, device=wd, generator=gen).clamp_min(1e-12)
    count = (10 + torch.floor(u.pow(-1.0 / 1.2) - 1.0)).clamp(max=1e5)     # 10 + floor(Pareto(1.2))
    return _finish(row, col, count, gen, device)


WORKLOADS = {
    # name: (generator, kwargs, d) — BASELINE.json configs[1], [2], [3] (per-GPU shard for [3])
    "text8_d64": ("text8_shaped", dict(V=10000), 64),
    "text8_v50k_d300": ("zipf_sampled", dict(V=50000, nnz=8_000_000), 300),
    "zipf_v400k_d300": ("zipf_sampled", dict(V=400000, nnz=25_000_000), 300),
    "zipf_v2m_d128": ("zipf_sampled", dict(V=2_000_000, nnz=25_000_000), 128),      # configs[4], per-GPU shard
}


def make_workload(name: str, seed=0, device="cpu", work_device=None):
    gen_name, kw, d = WORKLOADS[name]
    fn = text8_shaped if gen_name == "text8_shaped" else zipf_sampled
    row, col, w, y = fn(seed=seed, device=device, work_device=work_device, **kw)
    return dict(row=row, col=col, w=w, y=y, V=kw["V"], d=d, name=name)
