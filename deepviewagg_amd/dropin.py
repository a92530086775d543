"""Make the reference's plugin lookups resolve to the HIP-backed classes.

The reference finds its modules by name at the dotted paths
``torch_points3d.modules.multimodal.{pooling,fusion}`` (ModalityFactory.get_module,
models/base_architectures/unet.py:69-101), ``torch_points3d.core.multimodal.visibility``
(MapImages, core/data_transform/multimodal/image.py:214-215) and imports the data classes from
``torch_points3d.core.multimodal.{csr,image}``.  ``install()`` either patches an importable
``torch_points3d`` in place (attribute by attribute) or, when the package is absent, registers alias
modules under those dotted names in ``sys.modules``.
"""
import importlib
import sys
import types

_ALIASES = {
    "torch_points3d.modules.multimodal.pooling": "deepviewagg_amd.modules.multimodal.pooling",
    "torch_points3d.modules.multimodal.fusion": "deepviewagg_amd.modules.multimodal.fusion",
    "torch_points3d.modules.multimodal.dropout": "deepviewagg_amd.modules.multimodal.dropout",
    "torch_points3d.modules.multimodal.modules": "deepviewagg_amd.modules.multimodal.modules",
    "torch_points3d.core.data_transform.multimodal.image": "deepviewagg_amd.core.data_transform.multimodal.image",
    "torch_points3d.core.multimodal.csr": "deepviewagg_amd.core.multimodal.csr",
    "torch_points3d.core.multimodal.image": "deepviewagg_amd.core.multimodal.image",
    "torch_points3d.core.multimodal.visibility": "deepviewagg_amd.core.multimodal.visibility",
    "torch_points3d.utils.multimodal": "deepviewagg_amd.utils.multimodal",
    "torch_points3d.modules.SparseConv3d.modules": "deepviewagg_amd.modules.SparseConv3d.modules",
    "torch_points3d.modules.SparseConv3d.nn": "deepviewagg_amd.modules.SparseConv3d.nn",
}


def install(patch_existing=True):
    """Returns the list of dotted names that now resolve to deepviewagg_amd code."""
    done = []
    for ref_name, our_name in _ALIASES.items():
        ours = importlib.import_module(our_name)
        target = None
        if patch_existing:
            try:
                target = importlib.import_module(ref_name)
            except Exception:
                target = None
        if target is not None:
            for k, v in vars(ours).items():
                if not k.startswith("_"):
                    setattr(target, k, v)
        else:
            parts = ref_name.split(".")
            for i in range(1, len(parts)):
                pkg = ".".join(parts[:i])
                if pkg not in sys.modules:
                    m = types.ModuleType(pkg)
                    m.__path__ = []
                    sys.modules[pkg] = m
            sys.modules[ref_name] = ours
            setattr(sys.modules[".".join(parts[:-1])], parts[-1], ours)
        done.append(ref_name)
    return done
