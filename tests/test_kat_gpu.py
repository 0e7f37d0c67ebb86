"""The hand-derived known-answer tests of tests/test_oracle_kat.py — closed-form depths, exact coverage counts and
spans under the top-left rule, shading constants, blend and depth-write rules, sampler and mip behaviour, clipping
against closed forms — run against the HIP library DIRECTLY: none of these expectations comes out of the oracle, so
a misreading shared by oracle and kernels would still have to get past them.  (Parity proper, HIP == oracle bit for
bit, is tests/test_parity_gpu.py.)"""
import pytest

import test_oracle_kat as K

pytestmark = pytest.mark.gpu

ANALYTIC = [K.test_config1_colored_triangle, K.test_shading_constants, K.test_shared_edge_hit_once, K.test_fan_hit_once,
            K.test_reversed_z_and_ties, K.test_transparent_pass, K.test_sampler_behaviour, K.test_mip_chain,
            K.test_floor_clipped_both_ends, K.test_near_clip_wall_is_watertight, K.test_scissor_band_matches_full_frame,
            K.test_ragged_and_empty]


@pytest.mark.parametrize("kat", ANALYTIC, ids=lambda f: f.__name__[5:])
def test_known_answer_on_the_hip_library(hip, kat):
    kat(hip)


@pytest.mark.parametrize("d", [1.0, 10.0, 85.0, 1000.0])
def test_depth_table_on_the_hip_library(hip, d):
    K.test_depth_table(hip, d)
