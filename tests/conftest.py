import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference (build container only; skipped elsewhere)")


def load_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    return g, meta


def decoder_kwargs(profile):
    """EMS / T-EMS parameters of a golden profile, with the defaults of nbldpc_amd/profiles.py."""
    from nbldpc_amd.profiles import DEFAULTS
    p = dict(DEFAULTS)
    p.update(profile)
    return dict(ems_nm=p["ems_nm"], ems_nc=p["ems_nc"], ems_factor=p["ems_factor"], ems_offset=p["ems_offset"],
                tems_nr=p["tems_nr"], tems_nc=p["tems_nc"], tems_factor=p["tems_factor"], tems_offset=p["tems_offset"])


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle
