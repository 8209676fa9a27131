"""scarlet_amd -- MI355X-native proximal-gradient deblending engine behind scarlet's
Blend / Observation / Component API (reference: vineetbansal/scarlet, ``scarlet/__init__.py``).

Importing the package needs the built HIP library (``scarlet_amd/csrc/libscarlet_hip.so``);
computing anything needs a ROCm device.  There is no CPU fallback.
"""
from . import _lib                      # raises loudly when the HIP library is missing
from . import operator
from . import psf
from . import fft
from . import measurement
from . import interpolation
from . import update as _update_module
from .bbox import Box, trim, flux_at_edge
from .cache import Cache
from .update import (positive_sed, positive_morph, positive, normalized, sparse_l0, sparse_l1,
                     threshold, monotonic, translation, symmetric)
from .component import BlendFlag, Prior, Component, ComponentTree
from .source import (SourceInitError, get_pixel_sed, get_psf_sed, get_best_fit_seds,
                     build_detection_coadd, init_extended_source, init_combined_extended_source,
                     init_multicomponent_source, PointSource, CombinedExtendedSource,
                     ExtendedSource, MultiComponentSource, RandomSource)
from .observation import Frame, Observation
from .blend import Blend
from .batch import BlendBatch
from . import bbox, cache, component, source, observation, blend, batch, synth, distributed, io

update = _update_module
__version__ = "0.1.0"
