"""The synthetic weight / input recipe lives in tss-with-dprnn_amd/synthetic.py (it is plain numpy data generation, and
bench.py's product leg needs it); the tests and the oracle keep importing it under this name."""
from tss_with_dprnn_amd.synthetic import *  # noqa: F401,F403
from tss_with_dprnn_amd.synthetic import (SHIPPED_CFG, SHIPPED_SPE_EXTRA, full_cfg, make_mixture, make_sources,  # noqa: F401
                                          make_state, state_spec)
