"""The interleaved multi-GPU partition ("row_tile % G == r"; include/svr.h svr_set_row_interleave): a
context that owns every G-th 32-row tile row renders exactly those rows of the single-context frame — colour, depth
and the presented image, which is written in place and nowhere else — whatever G, whatever the scissor's origin, also
when the last tile row is cut short and when there are more ranks than tile rows.  Run on the oracle here and on the
HIP library under -m gpu (against its own full frame: full-frame parity HIP == oracle is tests/test_parity_gpu.py)."""
import numpy as np
import pytest

import svr_testlib as T

SENTINEL = 0xA5


def owned_rows(height, y0, stride, offset):
    y = np.arange(height)
    return ((y - y0) >= 0) & ((((y - y0) >> 5) % stride) == offset)


def render(lib, pkg, width, height, stride=1, offset=0, scissor=None, dst=None, queue_caps=None):
    """one frame of the small atrium; returns colour, depth, the presented image and the row costs' count"""
    r, scene, opaque, transparent = T.setup_sponza(lib, width, height, lod=8, tex_size=32)
    if queue_caps is not None:
        r.set_option(pkg.abi.OPT_QUEUE_CAPS, queue_caps)
    if scissor is not None:
        r.set_scissor(*scissor)
    r.set_row_interleave(stride, offset)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)
    if dst is not None:
        r.copy_to_swapchain(dst, width, height, pkg.abi.SWAPCHAIN_B8G8R8A8)
    r.sync()
    out = {"color": r.read_color(), "depth": r.read_depth(), "replayed": r.get_stats().replayed_passes}
    costs, y0, rows = r.row_costs()
    out["cost_rows"] = len(costs)
    r.close()
    return out


def host_image(width, height):
    img = np.full((height, width, 4), SENTINEL, np.uint8)
    return img, img.ctypes.data, lambda: img


def check_interleaved(lib, pkg, make_image, width, height, stride, scissor=None, queue_caps=None):
    y0 = scissor[1] if scissor else 0
    h_sc = scissor[3] if scissor else height
    inside = (np.arange(height) >= y0) & (np.arange(height) < y0 + h_sc)
    _, full_ptr, full_get = make_image(width, height)
    full = render(lib, pkg, width, height, scissor=scissor, dst=full_ptr)
    full_img = full_get().copy()
    seen = np.zeros(height, bool)
    for offset in range(stride):
        _, ptr, get = make_image(width, height)
        part = render(lib, pkg, width, height, stride, offset, scissor=scissor, dst=ptr, queue_caps=queue_caps)
        img = get()
        own = owned_rows(height, y0, stride, offset) & inside
        assert not (seen & own).any()
        seen |= own
        n_tile_rows = (h_sc + 31) // 32
        assert part["cost_rows"] == (max(0, n_tile_rows - offset) + stride - 1) // stride
        if not own.any():
            assert np.all(img == SENTINEL)
            continue
        T.assert_images_identical(part["color"][own], full["color"][own], f"colour rows of rank {offset}/{stride}")
        T.assert_images_identical(part["depth"][own], full["depth"][own], f"depth rows of rank {offset}/{stride}")
        T.assert_images_identical(img[own], full_img[own], f"presented rows of rank {offset}/{stride}")
        assert np.all(img[~own] == SENTINEL), "the present wrote rows the context does not own"
        if queue_caps is not None and lib.backend != "cpu-oracle":
            assert part["replayed"] >= 1
    assert np.array_equal(seen, inside)


@pytest.mark.parametrize("stride", [1, 2, 3, 8])
def test_interleaved_rows_compose_the_frame(oracle, pkg, stride):
    check_interleaved(oracle, pkg, host_image, 160, 200, stride)  # 200 rows: the last tile row is 8 rows tall


def test_interleaved_rows_under_a_scissor(oracle, pkg):
    check_interleaved(oracle, pkg, host_image, 160, 200, 3, scissor=(0, 20, 160, 150))  # tile rows count from row 20


def test_more_ranks_than_tile_rows(oracle, pkg):
    check_interleaved(oracle, pkg, host_image, 96, 70, 5)  # three tile rows, five ranks: two own nothing


def test_bad_arguments(oracle, pkg):
    r = oracle.create(64, 64)
    for stride, offset in ((0, 0), (65, 0), (4, 4)):
        with pytest.raises(pkg.abi.SvrError):
            r.set_row_interleave(stride, offset)
    r.close()


# ---------------------------------------------------------------- the HIP library
def device_image(width, height):
    import torch
    t = torch.full((height, width, 4), SENTINEL, dtype=torch.uint8, device="cuda")
    return t, t.data_ptr(), lambda: t.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [2, 3, 8])
def test_hip_interleaved_rows_compose_the_frame(hip, pkg, stride):
    check_interleaved(hip, pkg, device_image, 640, 360, stride)


@pytest.mark.gpu
def test_hip_interleaved_rows_under_a_scissor(hip, pkg):
    check_interleaved(hip, pkg, device_image, 640, 360, 3, scissor=(0, 20, 640, 300))


@pytest.mark.gpu
def test_hip_more_ranks_than_tile_rows(hip, pkg):
    check_interleaved(hip, pkg, device_image, 96, 70, 5)


@pytest.mark.gpu
def test_hip_interleaved_rows_survive_a_replay(hip, pkg):
    check_interleaved(hip, pkg, device_image, 640, 360, 4, queue_caps=64)


@pytest.mark.gpu
def test_hip_interleaved_rows_match_the_oracle(hip, oracle, pkg):
    a = render(hip, pkg, 640, 360, 4, 1)
    b = render(oracle, pkg, 640, 360, 4, 1)
    own = owned_rows(360, 0, 4, 1)
    T.assert_images_identical(a["color"][own], b["color"][own], "colour")
    T.assert_images_identical(a["depth"][own], b["depth"][own], "depth")


@pytest.mark.gpu
def test_hip_present_status_reports_a_void_present(hip, pkg):
    """svr_set_present_status: a present behind a pass that overflowed its queues is void and says so on the stream; the
    replay runs it again and takes the word back"""
    import torch
    w, h = 640, 360
    want_t, want_ptr, want_get = device_image(w, h)
    render(hip, pkg, w, h, dst=want_ptr)
    r, scene, opaque, transparent = T.setup_sponza(hip, w, h, lod=8, tex_size=32)
    status = torch.full((2,), 7, dtype=torch.int32, device="cuda")
    img, ptr, get = device_image(w, h)
    r.set_present_status(status.data_ptr())
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)
    r.copy_to_swapchain(ptr, w, h, pkg.abi.SWAPCHAIN_B8G8R8A8)
    torch.cuda.synchronize()
    assert status.tolist() == [0, 7] and np.array_equal(get(), want_get())
    r.set_option(pkg.abi.OPT_QUEUE_CAPS, 64)  # the next pass overflows: it and the present behind it are void
    img.fill_(SENTINEL)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)
    r.copy_to_swapchain(ptr, w, h, pkg.abi.SWAPCHAIN_B8G8R8A8)
    torch.cuda.synchronize()  # the device is idle, the library has not looked yet
    assert status.tolist() == [1, 7] and np.all(get() == SENTINEL)
    r.sync()                  # the fence finds the overflow and replays pass and present
    torch.cuda.synchronize()
    assert status.tolist() == [2, 7] and np.array_equal(get(), want_get())  # 2: carried out, but by the replay
    assert r.get_stats().replayed_passes >= 1
    r.set_present_status(0)
    r.close()
