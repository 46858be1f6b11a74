"""Known-answer tests for the fp64 SI-SDR / PIT restatement (oracle/sisdr.py; asteroid itself is absent)."""
import numpy as np

from oracle.sisdr import pit_si_sdr, si_sdr


def test_scale_invariance_and_known_value():
    rng = np.random.default_rng(0)
    t = rng.standard_normal(8000)
    n = rng.standard_normal(8000)
    n -= (n @ t) / (t @ t) * t                      # noise orthogonal to the target
    n *= np.linalg.norm(t) / np.linalg.norm(n) / 10  # |t|^2/|n|^2 = 100 -> 20 dB (means are ~0)
    t0, n0 = t - t.mean(), n - n.mean()
    want = 10 * np.log10((t0 @ t0) / (n0 @ n0))
    got = si_sdr(t + n, t)
    assert abs(got - 20.0) < 0.05 and abs(got - want) < 0.05
    assert abs(si_sdr(3.7 * (t + n), t) - got) < 1e-9            # scale invariant in the estimate
    assert abs(si_sdr(t + n + 5.0, t) - got) < 1e-9              # and offset invariant (zero-mean)


def test_pit_picks_the_right_permutation():
    rng = np.random.default_rng(1)
    s = rng.standard_normal((3, 2, 4000))
    est = s[:, ::-1] + 0.01 * rng.standard_normal((3, 2, 4000))   # swapped speakers
    score, perm = pit_si_sdr(est, s)
    assert (perm == np.array([1, 0])).all() and (score > 35).all()
