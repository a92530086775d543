"""-m gpu: ImageMapping rescale / crop / select_views and the online mapping transforms against vectors produced by
the REFERENCE's own source (tests/golden/transforms.npz, oracle/gen_golden.py::gen_transforms: the file
core/data_transform/multimodal/image.py is loaded directly, its package-level imports replaced by placeholders).
Integer outputs (pointers, image ids, pixel sets, rollings, crop offsets, picked images) are bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from test_gpu_data import mapping_equals, eq

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def fresh(g, dev=DEV):
    from deepviewagg_amd.core.multimodal.image import ImageMapping, SameSettingImageData
    N, B = int(g["num_points"]), g["x"].shape[0]
    m = ImageMapping.from_dense(t(g["point_ids"], dev), t(g["image_ids"], dev), t(g["pixels_dense"], dev),
                                t(g["map_features_dense"], dev), num_points=N)
    return SameSettingImageData(path=np.array([f"img_{i}" for i in range(B)]), pos=t(g["pos"], dev),
                                opk=torch.zeros(B, 3, device=dev), ref_size=tuple(int(v) for v in g["ref_size"]),
                                proj_upscale=1, mappings=m, x=t(g["x"], dev))


def test_fixture_input_mapping():
    g = load_golden("transforms")
    mapping_equals(fresh(g).mappings, g, "in_")


@pytest.mark.parametrize("r", [2, 4, 8])
def test_downscale_images_matches_reference(r):
    g = load_golden("transforms")
    mapping_equals(fresh(g).mappings.downscale_images(r), g, f"down{r}_")


def test_upscale_crop_select_views_bounding_boxes_match_reference():
    g = load_golden("transforms")
    m = fresh(g).mappings
    mapping_equals(m.upscale_images(2), g, "up2_")
    mapping_equals(m.upscale_images(3, center=False), g, "up3nc_")
    mapping_equals(m.crop(tuple(int(v) for v in g["crop_size"]), t(g["crop_offsets"], DEV)), g, "crop_")
    mv, seen = m.select_views(t(g["view_mask"], DEV))
    eq(seen, g["sv_seen_images"])
    mapping_equals(mv, g, "sv_")
    bb = torch.stack([b.long() for b in m.bounding_boxes], 0)
    eq(bb, g["bbox"])


def test_select_mapping_from_point_id_matches_reference():
    from deepviewagg_amd.core.data_transform.multimodal.image import SelectMappingFromPointId
    g = load_golden("transforms")
    keep = t(g["sel_keep"], DEV)
    data = Data(pos=torch.zeros(keep.shape[0], 3, device=DEV), mapping_index=keep.clone(), num_nodes=keep.shape[0])
    data, out = SelectMappingFromPointId()(data, fresh(g))
    eq(data.mapping_index, g["sel_mapping_index"])
    assert out.num_views == int(g["sel_num_views"])
    eq(out.pos, g["sel_pos"])
    mapping_equals(out.mappings, g, "sel_")


@pytest.mark.parametrize("tag,kw", [("area", dict(area_ratio=0.003, n_max=4)),
                                    ("bbox", dict(area_ratio=0.05, n_max=None, use_bbox=True))])
def test_pick_images_from_mapping_area_matches_reference(tag, kw):
    from deepviewagg_amd.core.data_transform.multimodal.image import PickImagesFromMappingArea
    g = load_golden("transforms")
    N = int(g["num_points"])
    data = Data(pos=torch.zeros(N, 3, device=DEV), mapping_index=torch.arange(N, device=DEV), num_nodes=N)
    _, out = PickImagesFromMappingArea(**kw)(data, fresh(g))
    eq(out.pos, g[f"pick_{tag}_pos"])
    eq(out.x, g[f"pick_{tag}_x"])
    mapping_equals(out.mappings, g, f"pick_{tag}_")


def _roll_crop(g):
    from deepviewagg_amd.core.data_transform.multimodal.image import CenterRoll, CropImageGroups
    N = int(g["num_points"])
    data = Data(pos=torch.zeros(N, 3, device=DEV), mapping_index=torch.arange(N, device=DEV), num_nodes=N)
    _, rolled = CenterRoll(angular_res=16)(data, fresh(g))
    return data, rolled


def test_center_roll_matches_reference():
    g = load_golden("transforms")
    _, rolled = _roll_crop(g)
    eq(rolled.rollings, g["roll_rollings"])
    eq(rolled.x, g["roll_x"])
    mapping_equals(rolled.mappings, g, "roll_")


def test_crop_image_groups_and_memory_credit_match_reference():
    from deepviewagg_amd.core.data_transform.multimodal.image import CropImageGroups, PickImagesFromMemoryCredit
    g = load_golden("transforms")
    data, rolled = _roll_crop(g)
    _, groups = CropImageGroups(padding=2, min_size=16)(data, rolled)
    assert len(groups) == int(g["crop_groups"])
    for gi, sd in enumerate(groups):
        assert tuple(sd.crop_size) == tuple(int(v) for v in g[f"cg{gi}_crop_size"])
        eq(sd.crop_offsets, g[f"cg{gi}_crop_offsets"])
        eq(sd.pos, g[f"cg{gi}_pos"])
        eq(sd.x, g[f"cg{gi}_x"])
        mapping_equals(sd.mappings, g, f"cg{gi}_")
    np.random.seed(int(g["credit_seed"]))
    _, picked = PickImagesFromMemoryCredit(credit=int(g["credit"]), k_coverage=2)(data, groups)
    assert len(picked) == int(g["credit_groups"])
    for gi, sd in enumerate(picked):
        assert tuple(sd.crop_size) == tuple(int(v) for v in g[f"mc{gi}_crop_size"])
        eq(sd.pos, g[f"mc{gi}_pos"])
        mapping_equals(sd.mappings, g, f"mc{gi}_")


def test_flip_and_pixel_feature_transforms_match_reference():
    from deepviewagg_amd.core.data_transform.multimodal.image import (RandomHorizontalFlip, ToFloatImage,
                                                                      AddPixelHeightFeature, AddPixelWidthFeature)
    g = load_golden("transforms")
    N = int(g["num_points"])
    data = Data(pos=torch.zeros(N, 3, device=DEV), mapping_index=torch.arange(N, device=DEV), num_nodes=N)
    sd = fresh(g)[torch.tensor([0, 3], device=DEV)]
    _, sd = RandomHorizontalFlip(p=1.0)(data, sd)
    eq(sd.x, g["flip_x"])
    mapping_equals(sd.mappings, g, "flip_")
    for tr in (ToFloatImage(), AddPixelHeightFeature(), AddPixelWidthFeature()):
        _, sd = tr(data, sd)
    np.testing.assert_allclose(sd.x[:, :, ::4, ::4].cpu().numpy(), g["feat_x"], rtol=0, atol=1e-7)
    # p = 0: nothing moves
    sd0 = fresh(g)
    before = sd0.mappings.pixels.clone()
    _, sd0 = RandomHorizontalFlip(p=0.0)(data, sd0)
    assert torch.equal(sd0.mappings.pixels, before)
