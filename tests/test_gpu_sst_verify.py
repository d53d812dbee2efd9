"""Whole-file table verification through the engine: same scenarios as the CPU tests of the host logic,
with the block checksums computed on the GPU."""
import pytest

import sst_cases
import sst_verify_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


@pytest.mark.parametrize("case", sst_cases.ALL, ids=lambda f: f.__name__)
def test_sst_verify_on_device(case, eng, tmp_path):
    case(U.verify_with_engine(eng), str(tmp_path))
