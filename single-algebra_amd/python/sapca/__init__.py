"""sapca: MI355X-native sparse PCA (the src/dimred/pca hot path of SingleRust/single-algebra).

Python is the test/bench host here; the product is libsapca.so (C ABI in include/sapca.h).
"""
from ._lib import LIB_PATH, SapcaError, load  # noqa: F401
from .pca import (DeviceCsr, MaskedSparsePCA, MaskedSparsePCABuilder, PowerIterationNormalizer,  # noqa: F401
                  SparsePCA, SparsePCABuilder, SVDMethod)
from .multi import MultiDevice  # noqa: F401
