"""CPU test of the drop-in seam: after install(), the reference's dotted paths resolve to our classes
(the lookups of unet.py:69-101 and core/data_transform/multimodal/image.py:214-215)."""
import importlib
import sys


def test_install_aliases_reference_paths():
    for k in [k for k in sys.modules if k.startswith("torch_points3d")]:
        del sys.modules[k]
    from deepviewagg_amd import dropin
    names = dropin.install(patch_existing=False)
    assert "torch_points3d.modules.multimodal.pooling" in names
    pooling = importlib.import_module("torch_points3d.modules.multimodal.pooling")
    for cls in ("BimodalCSRPool", "GroupBimodalCSRPool", "QKVBimodalCSRPool", "HeuristicBimodalCSRPool",
                "DeepSetFeat", "MinMaxDiffSetFeat", "MLPSetFeat", "Gating", "segment_softmax_csr",
                "gather_csr", "segment_gather_csr", "expand_group_feat", "nearest_power_of_2"):
        assert hasattr(pooling, cls), cls
    # ModalityFactory-style lookup + ctor with YAML-style kwargs (unknown kwargs are swallowed)
    view_pool = getattr(pooling, "GroupBimodalCSRPool")(
        in_map=8, in_mod=64, num_groups=4, use_mod=False, map_encoder="DeepSetFeat", use_num=True, index=0)
    keys = set(view_pool.state_dict())
    for k in ("E_map.mlp_elt_1.0.0.weight", "E_map.mlp_elt_1.0.1.batch_norm.running_mean",
              "E_map.mlp_set.1.1.batch_norm.weight", "E_mod.1.0.weight", "E_score.bias", "G.weight", "G.bias"):
        assert k in keys, k
    vis = importlib.import_module("torch_points3d.core.multimodal.visibility")
    model = getattr(vis, "SplattingVisibility")(img_size=(2048, 1024), voxel=0.02, r_max=8, r_min=0.05,
                                                k_swell=1.0, d_swell=1000, exact=True)
    assert "exact=True" in repr(model)
    import pickle
    assert pickle.loads(pickle.dumps(model)).voxel == 0.02          # stored on the image data, pickled
    fusion = importlib.import_module("torch_points3d.modules.multimodal.fusion")
    assert fusion.BimodalFusion(mode="concatenation").mode == "concatenation"
    # sparse-convolution stages and the backend shim (applications/sparseconv3d.py, SparseConv3d/nn/__init__.py)
    sp3d_nn = importlib.import_module("torch_points3d.modules.SparseConv3d.nn")
    sp3d_mod = importlib.import_module("torch_points3d.modules.SparseConv3d.modules")
    sp3d_nn.set_backend("torchsparse")
    assert sp3d_nn.get_backend() == "torchsparse" and sp3d_nn.backend_valid("minkowski")
    for name in ("cat", "Conv3d", "Conv3dTranspose", "ReLU", "SparseTensor", "BatchNorm"):
        assert hasattr(sp3d_nn, name), name
    stage = getattr(sp3d_mod, "ResNetDown")(down_conv_nn=[[16, 32]], kernel_size=2, stride=2, N=1, index=0)
    assert "blocks.0.block.0.kernel" in stage.state_dict() and "conv_in.1.bn.running_var" in stage.state_dict()
    assert getattr(sp3d_mod, "ResNetUp")(up_conv_nn=[32, 16, 16], N=1).CONVOLUTION == "Conv3dTranspose"
    for k in [k for k in sys.modules if k.startswith("torch_points3d")]:
        del sys.modules[k]
