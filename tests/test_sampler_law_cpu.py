"""The sampling law behind mk_exprace_topk's on-device generator, on the CPU (oracle/sampler_oracle.py): generating the race's
candidates by geometric skipping + thinning + conditional keys gives the same weighted sample without replacement as drawing
every Exp(1) -- inclusion counts, first-draw counts (two-sample chi-square), and the first draw ~ p / sum(p) (one-sample).
The kernel itself is pinned on the GPU (tests/test_solver_gpu.py: bit-exact selection with injected noise, chi-square of its own
draws against torch.multinomial)."""
import numpy as np

from oracle import sampler_oracle as S


def _chi2_two(a, b, min_count=16):
    sel = (a + b) >= min_count
    rest = (a[~sel].sum() - b[~sel].sum()) ** 2 / max(a[~sel].sum() + b[~sel].sum(), 1.0)
    return float((((a[sel] - b[sel]) ** 2) / (a[sel] + b[sel])).sum() + rest), int(sel.sum())


def test_skip_sampler_has_the_law_of_the_exponential_race():
    rng = np.random.default_rng(7)
    n, k, trials = 400, 24, 3000
    p = rng.random(n) ** 5 + 1e-3
    p[::9] = 0.0
    p[37] = 6.0                      # a dominant cell: its block goes the dense way
    p[200:216] *= 0.01               # a very light block
    T = S.threshold_for(p, 1.6 * k)
    inc_d, inc_s = np.zeros(n), np.zeros(n)
    first_d, first_s = np.zeros(n), np.zeros(n)
    short = 0
    for _ in range(trials):
        a = S.race_topk_direct(p, k, rng)
        b = S.race_topk_skip(p, k, T, rng)
        if b is None:                # fewer than k candidates (P ~ 1e-2 at 1.6 k expected): the kernel's `redo`
            short += 1
            b = S.race_topk_direct(p, k, rng)
        assert len(set(b.tolist())) == k and (p[b] > 0).all()
        np.add.at(inc_d, a, 1.0)
        np.add.at(inc_s, b, 1.0)
        first_d[a[0]] += 1
        first_s[b[0]] += 1
    assert short < 0.1 * trials
    for x, y in ((inc_d, inc_s), (first_d, first_s)):
        chi2, dof = _chi2_two(x, y)
        assert abs(chi2 - dof) < 5.0 * (2.0 * dof) ** 0.5, (chi2, dof)
    # the first draw of a race is one categorical draw ~ p
    exp = p / p.sum() * trials
    sel = exp >= 8
    chi2 = float((((first_s[sel] - exp[sel]) ** 2) / exp[sel]).sum() + (first_s[~sel].sum() - exp[~sel].sum()) ** 2 / exp[~sel].sum())
    dof = int(sel.sum())
    assert abs(chi2 - dof) < 5.0 * (2.0 * dof) ** 0.5, (chi2, dof)
    assert inc_s[::9].sum() == 0


def test_threshold_and_dense_branch_edges():
    rng = np.random.default_rng(1)
    p = np.full(64, 0.5)
    T = S.threshold_for(p, 40.0)
    assert np.sum(-np.expm1(-p / T)) >= 40.0 - 1e-6
    # every block dense (s = 0.9 per cell): the skip sampler is then a plain Bernoulli test of every cell
    got = [S.race_topk_skip(p, 8, T, rng) for _ in range(50)]
    assert all(g is not None and len(set(g.tolist())) == 8 for g in got)
    # fewer positive cells than k: never k candidates -> None (the caller's fallback pads, as the kernel's select does)
    q = np.zeros(64)
    q[:5] = 1.0
    assert S.race_topk_skip(q, 8, S.threshold_for(q, 4.0), rng) is None
