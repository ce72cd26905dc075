import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "separator_golden.npz")))


@pytest.fixture(scope="session")
def oracle_cfg_sd():
    from oracle import tfgridnet_oracle as O
    cfg = O.Cfg(**O.TSH_PARAMS)
    return cfg, O.synthetic_state_dict(cfg, seed=0)
