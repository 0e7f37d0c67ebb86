"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle):
the oracle must still reproduce them (CPU), and the HIP path must hit them bit for bit (GPU)."""
import os

import numpy as np
import pytest

import svr_testlib as T

import importlib.util
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(T.GOLDEN_DIR, "make_golden.py"))
MG = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MG)


def check(lib, name):
    gold = np.load(os.path.join(T.GOLDEN_DIR, name + ".npz"))
    assert str(gold["scene_sha256"]) == MG.scene_fingerprint(), \
        "the seeded scene generator no longer produces the bytes the goldens were made from"
    out = MG.CASES[name](lib)
    T.assert_images_identical(out["color"], gold["color"], name + " colour")
    T.assert_images_identical(out["depth"], gold["depth"], name + " depth")
    T.assert_images_identical(out["rgba8"], gold["rgba8"], name + " rgba8")


def check_mips(lib):
    gold = np.load(os.path.join(T.GOLDEN_DIR, "mips_64.npz"))
    tex, levels = MG.mip_case(lib)
    assert np.array_equal(tex, gold["level0"])
    for l in range(1, 7):
        assert np.array_equal(levels[l], gold[f"level{l}"]), f"mip level {l}"


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_oracle_reproduces_golden(oracle, name):
    check(oracle, name)


def test_oracle_reproduces_golden_mips(oracle):
    check_mips(oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_hip_matches_golden(hip, name):
    check(hip, name)


@pytest.mark.gpu
def test_hip_matches_golden_mips(hip):
    check_mips(hip)
