"""tests/test_eval_oracle.py (the evaluation arithmetic pinned to the reference's OWN outputs, tests/golden/eval_expected.npz) collected
a second time under the `gpu` marker, so that the GPU-box run executes it too (milliseconds, and the driver never ran it)."""
import pytest

from test_eval_oracle import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu
