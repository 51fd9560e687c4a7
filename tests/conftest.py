import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLD, name))


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def params(orc):
    return orc.Params()


@pytest.fixture(scope="session")
def gold_gate():
    return golden("gate_N1024.npz")


@pytest.fixture(scope="session")
def keys(orc, params, gold_gate):
    """The key set of the golden gate fixture, regenerated from its seed by the oracle's keygen."""
    k = orc.Keys(params, int(gold_gate["seed"]))
    assert orc.fnv64(k.bk_t) == int(gold_gate["bk_t_fnv"]), "oracle keygen drifted from the golden fixture"
    assert orc.fnv64(k.ksk) == int(gold_gate["ksk_fnv"])
    assert np.array_equal(k.key0, gold_gate["key0"]) and np.array_equal(k.key1, gold_gate["key1"])
    return k


@pytest.fixture(scope="session")
def engine(params, keys):
    """One GPU context with the golden key set loaded (torus BK -> spectra on the device)."""
    import rustfhe_amd as R
    p = R.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit)
    e = R.Engine(p, 0)
    e.load_bk_torus(keys.bk_t)
    e.load_ksk(keys.ksk)
    yield e
    e.close()
