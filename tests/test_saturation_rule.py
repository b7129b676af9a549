"""The order-free rule of the noisy-OR chain (mlm_sec_needs_order, mlmapping_amd/csrc/mlm_kernels_sector.h): a cell whose
contributions have strengths summing to >= 28 (strength 1/2/3 for a >= 0.5/0.75/0.875) ends at exactly 1.0f whatever the
order of the float chain p <- 1 - (1 - p)(1 - a) (update_odds_hashmap, map_awareness.h:147-154).  Checked here on random
multisets of odds (also adversarial ones: weakest strong values, many weak values in between) in random and sorted
orders, and that the threshold is not vacuous (just below it some order misses 1.0f)."""
import numpy as np

F = np.float32


def chain(a):
    p = F(0.0)
    one = F(1.0)
    for x in a:
        p = one - (one - p) * (one - x)
    return p


def strength(a):
    return np.where(a >= F(0.875), 3, np.where(a >= F(0.75), 2, np.where(a >= F(0.5), 1, 0)))


def test_strength_sum_28_saturates_in_any_order():
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(400):
        kind = trial % 4
        n_strong = int(rng.integers(10, 40))
        if kind == 0:  # the weakest values of every strength class
            strong = rng.choice(np.array([0.5, 0.75, 0.875], dtype=F), n_strong)
        elif kind == 1:  # just above the class borders
            strong = (rng.choice(np.array([0.5, 0.75, 0.875], dtype=F), n_strong) + rng.uniform(0, 1e-6, n_strong)).astype(F)
        else:
            strong = rng.uniform(0.5, 0.999, n_strong).astype(F)
        weak = rng.uniform(0.001, 0.4999, int(rng.integers(0, 200))).astype(F)
        a = np.concatenate([strong, weak]).astype(F)
        if strength(a).sum() < 28:
            continue
        checked += 1
        orders = [np.sort(a), np.sort(a)[::-1], np.concatenate([weak, strong]), np.concatenate([strong, weak])]
        orders += [rng.permutation(a) for _ in range(6)]
        for o in orders:
            assert chain(o) == F(1.0), (trial, strength(a).sum())
    assert checked > 150


def test_order_matters_below_the_threshold():
    """Why the other multi-kind cells are ranked: the float chain of weak contributions depends on their order, and strong
    steps that do not sum to the threshold do not reach 1.0f."""
    assert chain(np.full(23, 0.5, dtype=F)) != F(1.0)  # 23 halvings from p = 0: one grid step short
    rng = np.random.default_rng(9)
    found = 0
    for _ in range(200):
        a = rng.uniform(0.05, 0.45, 40).astype(F)
        if chain(np.sort(a)) != chain(np.sort(a)[::-1]):
            found += 1
    assert found > 20


def test_two_contributions_commute():
    """A cell with exactly two contributions needs no order (mlm_sec_needs_order): the first step sets p to the first value, the
    second is 1 - (1 - a)(1 - b) with a commutative float product — while three contributions already depend on their order."""
    rng = np.random.default_rng(11)
    vals = np.concatenate([rng.uniform(1e-6, 0.999999, 20000), [0.5, 0.75, 0.875, 1e-7, 0.9999999]]).astype(F)
    a, b = rng.permutation(vals), rng.permutation(vals)
    for x, y in zip(a, b):
        assert chain(np.array([x, y], dtype=F)) == chain(np.array([y, x], dtype=F))
    found = 0
    for _ in range(300):
        t = rng.uniform(0.05, 0.45, 3).astype(F)
        if len({float(chain(t[list(p)])) for p in ((0, 1, 2), (1, 2, 0), (2, 0, 1), (0, 2, 1))}) > 1:
            found += 1
    assert found > 10
