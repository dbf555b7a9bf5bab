#!/usr/bin/env python3
"""pytest under forced plan selectors: python3 tools/debug/pytest_tuned.py fx3_gen_tile=128 -- tests/test_hip_f16x3.py -m gpu -q"""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatiotemporalentropymodel_amd import _lib  # noqa: E402

i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    assert _lib.hip().stem_tuning_set(k.encode(), int(v)) == 0, _lib.hip().stem_last_error()
sys.exit(pytest.main(sys.argv[i + 1:]))
