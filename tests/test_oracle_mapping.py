"""The mapping-build oracle (oracle/mapping_oracle.c + .py) pinned against the reference's own outputs
(tests/golden/vis_*.npz, mapping_build.npz, lex_csr.npz — produced by oracle/gen_golden.py).

Bit-exact: every integer output (point indices, pixel coordinates, CSR pointers, image ids) and the
float32 depths.  Float projections of the equirectangular model agree to within the rounding of the
reference's float32 atan2/acos (NumPy's SIMD loops, <= 2 ulp of the angle); pinhole
and fisheye projections are bit-exact.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import mapping_oracle as M

VIS = ["vis_equirect_exact", "vis_equirect_dense", "vis_equirect_rot_crop_mask", "vis_equirect_bigsplat",
       "vis_pinhole_scannet", "vis_pinhole_kitti", "vis_fisheye_kitti", "vis_equirect_empty"]


def camera_of(g):
    kw = {k: g[k] for k in ("img_opk", "img_extrinsic", "img_intrinsic_pinhole", "img_intrinsic_fisheye") if k in g}
    return M.make_camera(str(g["camera"]), g["img_size"], g["img_xyz"], int(g["crop"][0]), int(g["crop"][1]),
                         float(g["r_min"]), float(g["r_max"]), float(g["voxel"]), float(g["k_swell"]),
                         float(g["d_swell"]), bool(g["exact"]), **kw)


@pytest.mark.parametrize("name", VIS)
def test_visibility_matches_reference(name):
    g = load_golden(name)
    cam = camera_of(g)
    mask = g.get("img_mask")
    i1, d, xp, yp = M.camera_projection(g["xyz"], cam, mask)
    assert np.array_equal(i1, g["proj_idx"])
    assert np.array_equal(d, g["proj_dist"])                      # float32 distances: bit-exact
    if "equirect" in name:
        # 2 ulp of a float32 angle in [-pi, pi] scaled to pixels: 2 * 2.4e-7 * W / (2 pi)
        tol = 2 * 2.4e-7 * max(g["img_size"]) / (2 * np.pi) * 1.5
        assert np.abs(xp - g["proj_x"]).max(initial=0) <= tol
        assert np.abs(yp - g["proj_y"]).max(initial=0) <= tol
    else:
        assert np.array_equal(xp, g["proj_x"]) and np.array_equal(yp, g["proj_y"])
    v = M.visibility(g["xyz"], cam, mask)
    for k in ("idx", "x", "y", "depth"):
        assert v[k].shape == g[k].shape and np.array_equal(v[k], g[k]), k
    if len(v["idx"]):
        f = M.mapping_features(g["xyz"], v, cam, g["linearity"], g["planarity"], g["scattering"], g["normals"])
        assert f.shape == g["features"].shape
        np.testing.assert_allclose(f, g["features"], rtol=0, atol=2.5e-7)


def test_map_images_assembly_matches_reference():
    """MapImages post-processing + ImageMapping.from_dense (image.py:238-417, core image.py:1728-1795)."""
    g = load_golden("mapping_build")
    proj = (int(g["ref_size"][0]) * int(g["proj_upscale"]), int(g["ref_size"][1]) * int(g["proj_upscale"]))
    cams = [M.make_camera("s3dis_equirectangular", proj, c, r_min=0.2, r_max=10.0, voxel=0.05, k_swell=1.0,
                          d_swell=1000, exact=True, img_opk=np.zeros(3)) for c in g["cams"]]
    seen, dense, mp = M.map_images(g["xyz"], cams, g["ref_size"], int(g["proj_upscale"]), g["linearity"],
                                   g["planarity"], g["scattering"], g["normals"])
    assert np.array_equal(seen, g["seen_images"])
    assert np.array_equal(dense["point_ids"], g["dense_point_ids"])
    assert np.array_equal(dense["image_ids"], g["dense_image_ids"])
    assert np.array_equal(dense["pixels"], g["dense_pixels"])
    np.testing.assert_allclose(dense["features"], g["dense_features"], rtol=0, atol=2.5e-7)
    for k in ("pointers", "images", "atom_pointers", "pixels"):
        assert mp[k].dtype == g[k].dtype and np.array_equal(mp[k], g[k]), k
    np.testing.assert_allclose(mp["features"], g["features"], rtol=0, atol=2.5e-7)


def test_lex_and_csr_match_reference():
    g = load_golden("lex_csr")
    assert np.array_equal(M.composite(g["a"], g["b"], g["c"]), g["composite"])
    assert np.array_equal(M.lexargunique(g["a"], g["b"], g["c"]), g["argunique"])
    u = M.lexargunique(g["a"], g["b"], g["c"])
    assert np.array_equal(g["a"][u], g["unique_a"]) and np.array_equal(g["c"][u], g["unique_c"])
    order = M.lexargsort_stable(g["a"], g["b"], g["c"])
    assert np.array_equal(g["composite"][order], g["argsort_keys"])
    assert np.array_equal(M.sorted_to_pointers(g["csr_idx"]), g["csr_pointers"])
