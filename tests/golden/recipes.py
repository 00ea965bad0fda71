"""Fixture-side name of the deterministic recipes: the functions live in ``torch_mnf_amd.synthetic`` (the
benchmark uses them too, and the benchmark must not depend on the tests tree)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from torch_mnf_amd.synthetic import *  # noqa: E402,F401,F403
from torch_mnf_amd.synthetic import _linear  # noqa: E402,F401
