"""GPU: the end-to-end example (render -> resample -> camera render pipeline -> losses -> FusedAdam -> prune, all on the
HIP library) optimises."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_synthetic_training_loss_falls():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import train_synthetic

    first, last, n = train_synthetic.main(["--gaussians", "30000", "--size", "192", "--iters", "120", "--quiet"])
    assert last < 0.6 * first, (first, last)
    assert 0 < n <= 30000
