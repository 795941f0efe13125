"""HIP path against the committed golden fixtures (tests/golden/*.npz)."""
import numpy as np
import pytest

from tests import golden_io

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("name", golden_io.names())
def test_carve_matches_golden(arvx, name, flags):
    g = golden_io.load(name)
    with arvx.Context(g["X"], g["Y"], g["Z"], g["s"]) as ctx:
        ctx.set_views(g["M"], g["masks"])
        ctx.carve(flags)
        assert np.array_equal(ctx.download_state(), g["state"])
