import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_cpu.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_raster():
    """reference ``fit`` run on the oracle rasteriser through the pytorch3d stubs (make_golden_raster.py)"""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_raster_cpu.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def smpl_struct():
    from mhhip import synthetic
    return synthetic.make_smpl_struct(1)


@pytest.fixture(scope='session')
def smpl_regs(smpl_struct):
    from mhhip import synthetic
    return synthetic.make_extra_regressors(1, smpl_struct)


@pytest.fixture(scope='session')
def oracle_model(smpl_struct, smpl_regs):
    from oracle import lbs_oracle
    return lbs_oracle.BodyModel(smpl_struct, smpl_regs)
