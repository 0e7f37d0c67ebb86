"""tests/test_spv_exec.py's checks — the reference's committed SPIR-V modules, executed, against the shader stages —
run on the HIP library: its vertex programs through svr_run_mesh_vert / svr_run_vertex_shader, its fragment programs
through whole passes and the pixel trace of the tile kernel, its compute programs through svr_draw_background.  No
oracle in the loop: the expectations are the fixture's (tests/golden/spv_exec.npz)."""
import numpy as np
import pytest

import test_spv_exec as X

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(X.FIX))


CHECKS = [X.check_mesh_vert, X.check_tex_image_vert, X.check_colored_triangle, X.check_mesh_frag, X.check_tex_image_frag,
          X.check_gradient, X.check_sky]


@pytest.mark.parametrize("check", CHECKS, ids=lambda f: f.__name__[6:])
def test_executed_module_on_the_hip_library(hip, pkg, fx, check):
    check(hip, pkg, fx)
