import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the library honours its test switches (LVA_WORK_CAP) only together with this one (environment switches)
os.environ["LVA_TESTING"] = "1"


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="run the minutes-long oracle cases")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU oracle case")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Its decodes are remembered for the session: the kernel modes are tested on the
    same seeded reads, and the oracle -- not the GPU -- is what the GPU suite's wall time is made of."""
    import hashlib
    from oracle import oracle as O
    O.lib()
    if not getattr(O.OracleCode, "_memo", None):
        memo = {}
        plain = O.OracleCode.decode
        plain_init = O.OracleCode.__init__

        def init(self, mem_conv, rate, msg_len, rc=False, sync_marker="", sync_period=0):
            plain_init(self, mem_conv, rate, msg_len, rc, sync_marker, sync_period)
            self._key = (mem_conv, rate, msg_len, bool(rc), sync_marker or "", sync_period)

        def decode(self, post, list_size, max_deviation=None, num_threads=1, max_steps=0, band_fma=True):
            import numpy as np
            pb = np.ascontiguousarray(post, dtype=np.float32)
            key = (self._key, list_size, max_deviation, num_threads, max_steps, band_fma, pb.shape, hashlib.sha1(pb.tobytes()).digest())
            if key not in memo:
                memo[key] = plain(self, post, list_size, max_deviation, num_threads, max_steps, band_fma)
            m, s = memo[key]
            return m.copy(), s.copy()

        O.OracleCode.__init__, O.OracleCode.decode, O.OracleCode._memo = init, decode, memo
    return O
