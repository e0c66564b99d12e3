"""tests/test_wrapper_pinned.py (keypoint attach, IoU, summary texts and the model table pinned to the reference's OWN outputs,
tests/golden/wrapper_expected.json) collected a second time under the `gpu` marker, so that the GPU-box run executes it too."""
import pytest

from test_wrapper_pinned import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu
