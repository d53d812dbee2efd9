"""Host logic of the whole-file table verification (footer, metaindex, index walking, extents, status
texts) with the checksum arithmetic supplied by the CPU oracle through the callback seam."""
import pytest

import sst_cases
import sst_verify_util as U


@pytest.mark.parametrize("case", sst_cases.ALL, ids=lambda f: f.__name__)
def test_sst_verify_host_logic(case, tmp_path):
    case(U.verify_with_oracle, str(tmp_path))
