"""Tile lists of the HIP path against the oracle's binning (reference: renderer.cu:165-219, one global stable radix
sort of (tile | depth bits) keys): every route of dm2_binning.hip must leave the same face_list and ranges --
per-tile sorts in LDS (by counting up to 512 entries, a bitonic network up to 2048), per-tile sorts in global memory (up to 32768), the radix route
(longer lists, DM2_FLAG_LEGACY_KERNELS) -- including the order of entries with EQUAL depth keys (emission order =
face id ascending), which only a stable sort or a (depth, face id) key reproduces."""
import numpy as np
import pytest
import torch

from util import capture_forward_args, scenes

pytestmark = pytest.mark.gpu


def _lists(args, legacy=False):
    from dmesh2_renderer_amd import _C
    dargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
    old = _C.set_flags(_C.DM2_FLAG_LEGACY_KERNELS if legacy else 0)
    try:
        out = _C.render_forward_cuda(*dargs)
        torch.cuda.synchronize()
    finally:
        _C.set_flags(old)
    R = out[0]
    B, H, W = out[2].shape
    N, Tn = B * H * W, B * ((W + 15) // 16) * ((H + 15) // 16)
    ranges = _C.debug_fetch(0, N, Tn, R, out[9], torch.int32, Tn * 2).cpu().numpy().view(np.uint32).reshape(Tn, 2)
    flist = _C.debug_fetch(1, N, Tn, R, out[8], torch.int32, R).cpu().numpy().view(np.uint32) if R else np.zeros(0, np.uint32)
    return R, ranges, flist


def _oracle_lists(args):
    from oracle import cpu as orc
    a = [x.numpy() if torch.is_tensor(x) else x for x in args]
    B, P = a[8].shape[0], a[4].shape[0]
    F, W, H = a[5].shape[0], int(a[2]), int(a[3])
    b = orc.Binning(B, P, F, W, H, a[1], a[5], a[8], a[9])
    return b.num_rendered, b.ranges, b.face_list


def _scene(W, H, F, seed, dc, cams=1, dup=False):
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + 700 + seed, num_cams=cams, shared_verts=False, depth_complexity=dc)
    if dup:     # every face four times: equal depth keys, the order inside a tie is the face id
        sc.faces = sc.faces.repeat(4, 1)
        sc.faces_opacity = sc.faces_opacity.repeat(4)
        sc.faces_intense = sc.faces_intense.repeat(1, 4)
    bidx = list(range(cams))
    return capture_forward_args(sc, bidx, [[0, 0]] * cams, W, H, 0.0, 0)[0]


CASES = {
    # name: (W, H, F, depth complexity, cameras, duplicated faces, expected longest list at least)
    "short_lists": (96, 80, 4000, 6.0, 2, False, 1),               # ordered by counting (<= 512 entries)
    "lds_network": (64, 48, 6000, 60.0, 1, False, 513),             # bitonic network in LDS (513 .. 2048)
    "ties": (64, 64, 700, 8.0, 1, True, 1),
    "one_tile_global_sort": (16, 16, 3000, 3000.0, 1, False, 2049),
    "mixed_lds_global": (48, 32, 9000, 2500.0, 1, False, 2049),
    "radix_fallback": (16, 16, 40000, 40000.0, 1, False, 32769),
}


@pytest.mark.parametrize("case", list(CASES))
def test_tile_lists_match_oracle(case):
    W, H, F, dc, cams, dup, longest_min = CASES[case]
    args = _scene(W, H, F, len(case), dc, cams, dup)
    R_ref, ranges_ref, flist_ref = _oracle_lists(args)
    longest = int((ranges_ref[:, 1] - ranges_ref[:, 0]).max()) if R_ref else 0
    assert longest >= longest_min, f"the scene does not reach the route it is meant for (longest list {longest})"
    for legacy in (False, True):
        R, ranges, flist = _lists(args, legacy)
        assert R == R_ref
        assert np.array_equal(ranges, ranges_ref), f"{case} legacy={legacy}: ranges differ"
        assert np.array_equal(flist, flist_ref), f"{case} legacy={legacy}: {int((flist != flist_ref).sum())} list entries differ"


def test_empty_and_untouched_tiles():
    """No face reaches a tile: num_rendered 0, all ranges (0, 0); a scene that leaves most tiles empty keeps (0, 0) there."""
    from dmesh2_renderer_amd import _C
    args = list(_scene(64, 48, 4, 3, 0.02))
    R_ref, ranges_ref, flist_ref = _oracle_lists(args)
    R, ranges, flist = _lists(args)
    assert R == R_ref and np.array_equal(ranges, ranges_ref) and np.array_equal(flist, flist_ref)
    assert (ranges_ref[:, 1] == ranges_ref[:, 0]).any(), "expected some empty tiles"
    far = [a.clone() if torch.is_tensor(a) else a for a in args]
    far[8] = far[8] + 10.0          # every vertex behind the far plane: culled (forward.cu:71)
    R0, ranges0, _ = _lists(far)
    assert R0 == 0 and not ranges0.any()


def test_small_and_large_faces_share_tiles():
    """Faces of at most four tiles take their place in the tile segment during the plan, larger ones are counted apart and
    placed behind them through a cursor (dm2_binning.hip): a scene that mixes both in every tile, two views."""
    W, H, cams = 128, 96, 2
    small = scenes.triangle_soup(W, H, 3000, scenes.SEED_BASE + 811, num_cams=cams, shared_verts=False, depth_complexity=3.0)
    large = scenes.triangle_soup(W, H, 60, scenes.SEED_BASE + 812, num_cams=cams, shared_verts=False, depth_complexity=40.0)
    sc = small
    P0 = small.verts.shape[0]
    sc.verts = torch.cat([small.verts, large.verts]); sc.verts_color = torch.cat([small.verts_color, large.verts_color])
    # interleave: the large faces sit between the small ones in face order
    f_all = torch.cat([small.faces, large.faces + P0]); perm = torch.randperm(f_all.shape[0], generator=torch.Generator().manual_seed(3))
    sc.faces = f_all[perm].contiguous()
    sc.faces_opacity = torch.cat([small.faces_opacity, large.faces_opacity])[perm].contiguous()
    sc.faces_intense = torch.cat([small.faces_intense, large.faces_intense], dim=1)[:, perm].contiguous()
    args = capture_forward_args(sc, [0, 1], [[0, 0]] * 2, W, H, 0.0, 0)[0]
    R_ref, ranges_ref, flist_ref = _oracle_lists(args)
    from oracle import cpu as orc
    a = [x.numpy() if torch.is_tensor(x) else x for x in args]
    tt = orc.Binning(a[8].shape[0], a[4].shape[0], a[5].shape[0], W, H, a[1], a[5], a[8], a[9]).tiles_touched
    assert (tt > 4).sum() >= 40 and ((tt > 0) & (tt <= 4)).sum() >= 2000, "the scene must mix both kinds of faces"
    for legacy in (False, True):
        R, ranges, flist = _lists(args, legacy)
        assert R == R_ref and np.array_equal(ranges, ranges_ref) and np.array_equal(flist, flist_ref), legacy
