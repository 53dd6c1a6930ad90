"""Host-side mirror of the R API: parameter checks, recycling and shapes (LogitWrapper.R), no GPU needed."""
import numpy as np

import bayeslogit_amd as bl
from bayeslogit_amd import api
from bayeslogit_amd.dist import shard_range


def test_parameter_checks_print_and_return_na(capsys):
    assert bl.rpg(5, h=0.0) is None                      # LogitWrapper.R:107-110
    assert "h must be > 0." in capsys.readouterr().out
    assert bl.rpg_devroye(5, n=-1) is None               # :37-40
    assert bl.rpg_alt(5, h=0.5) is None                  # :56-59
    assert "h must be >= 1." in capsys.readouterr().out
    assert bl.rpg_sp(5, h=0.99) is None                  # :77-80
    assert bl.rpg_gamma(5, h=-0.1) is None               # :15-18
    assert bl.rpg_gamma(5, h=1.0, trunc=0) is None       # :19-22
    assert "trunc must be > 0." in capsys.readouterr().out


def test_check_parameters(capsys):
    y = np.array([0.0, 1.0, 0.5])
    n = np.ones(3)
    ok = api._check_parameters(y, n, np.zeros(2), np.zeros((2, 2)), 3, 2, 10, 0)
    assert ok
    assert not api._check_parameters(np.array([0.0, 1.5, 0.5]), n, np.zeros(2), np.zeros((2, 2)), 3, 2, 10, 0)
    assert "y is a proportion" in capsys.readouterr().out
    assert not api._check_parameters(y, np.array([1.0, 0.0, 1.0]), np.zeros(2), np.zeros((2, 2)), 3, 2, 10, 0)
    assert not api._check_parameters(y, n, np.zeros(3), np.zeros((2, 2)), 3, 2, 10, 0)
    assert not api._check_parameters(y, n, np.zeros(2), np.zeros((3, 3)), 3, 2, 10, 0)
    assert not api._check_parameters(y, n, np.zeros(2), np.zeros((2, 2)), 3, 2, 0, 0)
    assert not api._check_parameters(y, n, np.zeros(2), np.zeros((2, 2)), 3, 2, 1, -1)
    assert bl.logit_combine(np.array([0.0, 2.0]), np.eye(2)) == -1          # LogitWrapper.R:171-172
    assert bl.mlogit_combine(np.array([[0.7, 0.7]]), np.ones((1, 1))) is None


def test_recycle_matches_R_array():
    assert api._recycle([1.0, 2.0], 5).tolist() == [1, 2, 1, 2, 1]           # array(h, num)
    assert api._recycle(3.0, 3).tolist() == [3, 3, 3]
    assert api._recycle([1, 2, 3, 4], 2, np.int32).tolist() == [1, 2]


def test_shard_range_partitions():
    for N in (0, 1, 7, 10_000_001):
        for W in (1, 2, 3, 8):
            parts = [shard_range(N, r, W) for r in range(W)]
            assert parts[0][0] == 0 and parts[-1][1] == N
            assert all(parts[i][1] == parts[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
