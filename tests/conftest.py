import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Fixtures captured from the reference's Python (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture()
def oracle_backend():
    """CPU tests of HOST logic only: for the duration of a test the oracle is registered as the CPU kernel of the
    rigid-transform dispatcher ops (the product registers none: it has no CPU path).  The registration is dropped
    again when the test ends."""
    from oracle import transform_convert as o

    import nesvor_amd.ops  # noqa: F401

    lib = torch.library.Library("nesvor", "IMPL")
    lib.impl("axisangle2mat_forward", o.axisangle2mat_forward, "CPU")
    lib.impl("axisangle2mat_backward", o.axisangle2mat_backward, "CPU")
    lib.impl("mat2axisangle_forward", o.mat2axisangle_forward, "CPU")
    lib.impl("mat2axisangle_backward", o.mat2axisangle_backward, "CPU")
    yield lib
    lib._destroy()


def small_args(**over):
    from argparse import Namespace

    a = Namespace(
        n_features_per_level=2, log2_hashmap_size=12, level_scale=1.3819, coarsest_resolution=16.0,
        finest_resolution=2.0, n_levels_bias=0, depth=1, width=64, n_features_z=15, n_features_slice=16,
        no_transformation_optimization=False, no_slice_scale=False, no_pixel_variance=False,
        no_slice_variance=False, single_precision=True, weight_transformation=0.1, weight_bias=100.0,
        image_regularization="edge", weight_image=2.0, delta=0.2, learning_rate=5e-3, gamma=0.33,
        milestones=[0.5, 0.75, 0.9], n_iter=20, batch_size=64, n_samples=8, output_resolution=2.0,
        output_intensity_mean=700.0, mask_threshold=1.0, no_output_psf=False, debug=False,
        device=torch.device("cpu"), dtype=torch.float32,
    )
    for k, v in over.items():
        setattr(a, k, v)
    a.inference_batch_size = 8 * a.batch_size
    a.n_inference_samples = 2 * a.n_samples
    return a


TRANSFORM_TABLE = [
    [0, 0, 0, 0, 0, 0],
    [np.pi / 2, 0, 0, 1, 2, 3],
    [0, -np.pi / 2, 0, -1.1, -10, 100.5],
    [0, 0, np.pi - 0.01, 2, 1, 10.5],
    [0, -np.pi + 0.01, 0, 2, 1, 10.5],
    [0.1, 0.1, 0.1, 0.1, 0.1, 0.1],
    [-0.1, 0, -0.4, 0.1, 0.5, 0.1],
    [-0.2, 0.2, -0.1, -100, 200, -159],
    [-0.12, -0.01, 0.1, -100, 200, -159],
    [np.pi / 4, np.pi / 4, np.pi / 4, 0.1, 0.1, 0.1],
    [np.pi / 3, -np.pi / 4, np.pi / 5, 100, 200, -300],
]


def scipy_table():
    """The reference's own known-answer table (tests/__init__.py:18-36): rotvec -> scipy matrix."""
    from scipy.spatial.transform import Rotation

    ax = torch.tensor(TRANSFORM_TABLE, dtype=torch.float32)
    R = torch.tensor(Rotation.from_rotvec(ax[:, :3].numpy().astype(np.float64)).as_matrix(), dtype=torch.float32)
    return ax, torch.cat([R, ax[:, 3:, None]], -1)
