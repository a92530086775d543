"""The two visibility models that decide per projected point -- DepthBasedVisibility, BiasuttiVisibility (reference
core/multimodal/visibility.py:1356-1496, :1779-1803) -- against fixtures written by the reference's own classes
(oracle/gen_golden.py visibility_models: KeOps through the brute-force shim, the S3DIS depth PNG through PIL).
CPU: the oracle restatement (oracle/visibility_models_oracle.py) on the fixtures' projections.
GPU: the classes of deepviewagg_amd.core.multimodal.visibility (dva_camera_projection + dva_knn + dva_mapping_features)."""
import os

import numpy as np
import pytest
import torch

from oracle import visibility_models_oracle as VO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BIAS = ["vis_biasutti", "vis_biasutti_wrap"]


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def _sub(proj_idx, idx):
    """positions of `idx` (a subset, in order) inside `proj_idx`"""
    pos = {int(v): i for i, v in enumerate(proj_idx)}
    return np.array([pos[int(v)] for v in idx], dtype=np.int64)


@pytest.mark.parametrize("name", BIAS)
def test_oracle_biasutti_matches_reference(name):
    g = load(name)
    k, margin, thr = int(g["k"]), int(g["margin"]), float(g["threshold"])
    nbr = VO.knn_image(g["proj_x"], g["proj_y"], k, None if margin < 0 else margin, int(g["img_size"][0]))
    assert np.array_equal(nbr, g["neighbors"])
    keep, _ = VO.biasutti_visibility(g["proj_x"], g["proj_y"], g["proj_dist"], g["img_size"], k=k,
                                     margin=None if margin < 0 else margin, threshold=None if thr < 0 else thr)
    assert np.array_equal(g["proj_idx"][keep], g["idx"])
    assert np.array_equal(g["proj_x"][keep], g["x"]) and np.array_equal(g["proj_y"][keep], g["y"])


def test_oracle_depth_map_matches_reference():
    g = load("vis_depth_map")
    # the S3DIS format (visibility.py:1326-1355): 1/512 m, 2^16 - 1 = missing -> -1
    dm = g["depth_png_u16"].astype(np.float32) / 512
    dm[g["depth_png_u16"] == 65535] = -1
    assert np.array_equal(dm, g["depth_map"])
    keep = VO.depth_map_visibility(g["proj_x"], g["proj_y"], g["proj_dist"], g["depth_map"], float(g["depth_threshold"]))
    assert np.array_equal(g["proj_idx"][keep], g["idx"])
    assert 0 < len(keep) < len(g["proj_idx"])


# ---------------------------------------------------------------------------------------------------------------------
DEV = "cuda:0"


def _call(model, g, **extra):
    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return model(t(g["xyz"]), t(g["img_xyz"]), linearity=t(g["linearity"]), planarity=t(g["planarity"]),
                 scattering=t(g["scattering"]), normals=t(g["normals"]), img_opk=t(g["img_opk"]), **extra)


def _base(g):
    return dict(img_size=tuple(int(v) for v in g["img_size"]), crop_top=0, crop_bottom=0, r_max=float(g["r_max"]),
                r_min=float(g["r_min"]), camera="s3dis_equirectangular")


def _check(out, g):
    assert np.array_equal(out["idx"].cpu().numpy(), g["idx"])
    assert np.array_equal(out["depth"].cpu().numpy(), g["depth"])          # float32 distances: bit-exact
    # float projections of the equirectangular model: within the angle rounding of the float-width contract
    np.testing.assert_allclose(out["x"].cpu().numpy(), g["x"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["y"].cpu().numpy(), g["y"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["features"].cpu().numpy(), g["features"], rtol=0, atol=2.5e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", BIAS)
def test_gpu_biasutti_matches_reference(name):
    from deepviewagg_amd.core.multimodal.visibility import BiasuttiVisibility
    g = load(name)
    k, margin, thr = int(g["k"]), int(g["margin"]), float(g["threshold"])
    model = BiasuttiVisibility(k=k, margin=None if margin < 0 else margin, threshold=None if thr < 0 else thr, **_base(g))
    # the neighbour search alone, on the reference's projections: exact (fp32 distances, ties to the lower index)
    nbr = model._neighbors(torch.from_numpy(g["proj_x"]).to(DEV), torch.from_numpy(g["proj_y"]).to(DEV))
    assert np.array_equal(nbr.cpu().numpy(), g["neighbors"])
    _check(_call(model, g), g)


@pytest.mark.gpu
def test_gpu_biasutti_default_k_and_batch():
    """k = 75 (the reference's default, beyond the 64 of the neighbourhood features) against the oracle, and the batch
    contract MapImages calls."""
    from deepviewagg_amd.core.multimodal.visibility import BiasuttiVisibility
    g = load("vis_biasutti")
    model = BiasuttiVisibility(**_base(g))
    assert model.k == 75
    out = _call(model, g)
    keep, _ = VO.biasutti_visibility(g["proj_x"], g["proj_y"], g["proj_dist"], g["img_size"], k=75)
    assert np.array_equal(out["idx"].cpu().numpy(), g["proj_idx"][keep])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)       # noqa: E731
    b = model.batch(t(g["xyz"]), torch.stack([t(g["img_xyz"])] * 2), img_opk=torch.stack([t(g["img_opk"])] * 2))
    n = out["idx"].shape[0]
    assert b["row_ptr"].tolist() == [0, n, 2 * n] and torch.equal(b["idx"][:n], out["idx"]) \
        and torch.equal(b["idx"][n:], out["idx"]) and b["image"].tolist() == [0] * n + [1] * n


@pytest.mark.gpu
def test_gpu_depth_based_matches_reference(tmp_path):
    from PIL import Image
    from deepviewagg_amd.core.multimodal.visibility import DepthBasedVisibility
    g = load("vis_depth_map")
    model = DepthBasedVisibility(depth_threshold=float(g["depth_threshold"]), **_base(g))
    _check(_call(model, g, depth_map=torch.from_numpy(g["depth_map"])), g)          # the loaded map
    path = str(tmp_path / "depth.png")
    Image.fromarray(g["depth_png_u16"].T).save(path)                                 # the S3DIS file format
    _check(_call(model, g, depth_map_path=path), g)
    with pytest.raises(AssertionError):
        _call(model, g)                                                              # visibility.py:1374


@pytest.mark.gpu
def test_gpu_map_images_with_depth_based_visibility(tmp_path):
    """MapImages(method='DepthBasedVisibility') end to end: the per-image depth file is derived from image.path
    as the reference does (core/data_transform/multimodal/image.py:262-265: '<area>/depth/<name>_depth.png' next to
    '<area>/<dir>/<name>_rgb.png') and reaches the model image by image through VisibilityModel.batch.  Two images at the
    same pose with DIFFERENT depth files: the first sees what the fixture's reference run kept, the second (all depths
    'missing' = 65535 -> -1) keeps nothing and is dropped."""
    from types import SimpleNamespace
    from PIL import Image
    from deepviewagg_amd.core.data_transform.multimodal import MapImages
    from deepviewagg_amd.core.multimodal.image import SameSettingImageData
    g = load("vis_depth_map")
    area = tmp_path / "Area_1"
    (area / "data").mkdir(parents=True)
    (area / "depth").mkdir()
    Image.fromarray(g["depth_png_u16"].T).save(str(area / "depth" / "cam_a_depth.png"))
    Image.fromarray(np.full_like(g["depth_png_u16"], 65535).T).save(str(area / "depth" / "cam_b_depth.png"))
    paths = np.array([str(area / "data" / "cam_a_rgb.png"), str(area / "data" / "cam_b_rgb.png")])
    n = g["xyz"].shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))       # noqa: E731
    data = SimpleNamespace(pos=t(g["xyz"]), mapping_index=torch.arange(n), linearity=t(g["linearity"]),
                           planarity=t(g["planarity"]), scattering=t(g["scattering"]), norm=t(g["normals"]))
    W, H = (int(v) for v in g["img_size"])
    images = SameSettingImageData(path=paths, pos=torch.stack([t(g["img_xyz"])] * 2), opk=torch.stack([t(g["img_opk"])] * 2),
                                  ref_size=(W, H), proj_upscale=1)
    tr = MapImages(method="DepthBasedVisibility", r_max=float(g["r_max"]), r_min=float(g["r_min"]),
                   depth_threshold=float(g["depth_threshold"]), camera="s3dis_equirectangular", crop_top=0, crop_bottom=0)
    _, out = tr(data, images)
    assert out.num_views == 1 and str(out.path[0]).endswith("cam_a_rgb.png")      # the second image sees nothing
    m = out.mappings
    # every point the reference's model kept is mapped (first occurrence per (point, pixel): a subset of the kept rows)
    kept = set(np.unique(g["idx"]).tolist())
    mapped = set(torch.nonzero(m.pointers[1:] > m.pointers[:-1]).view(-1).tolist())
    assert mapped and mapped <= kept
    # pixels: floor of the reference's float projections of the kept points
    px = {int(i): (int(x), int(y)) for i, x, y in zip(g["idx"], g["x"], g["y"])}
    pts = torch.repeat_interleave(torch.arange(m.num_groups), m.pointers[1:] - m.pointers[:-1])
    pix = m.pixels
    assert pix.shape[0] == pts.shape[0]
    bad = sum(1 for p, (x, y) in zip(pts.tolist(), pix.tolist()) if abs(px[p][0] - x) > 1 or abs(px[p][1] - y) > 1)
    assert bad == 0
