from .csr import CSRData, CSRBatch  # noqa: F401
from .image import (SameSettingImageData, SameSettingImageBatch, ImageData, ImageBatch,  # noqa: F401
                    ImageMapping, ImageMappingBatch, sparse_interpolation)
from .visibility import VisibilityModel, SplattingVisibility  # noqa: F401
