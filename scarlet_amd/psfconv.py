"""Device-side PSF convolution used by Observation.render (row a3b): the reference's
zero-pad -> ifftshift -> rfftn -> * K-hat -> irfftn -> fftshift -> centre-crop chain
(reference observation.py:198-201, fft.py:304-317) as batched hipFFT transforms plus index-
mapping kernels in the HIP library."""
import numpy as np

from . import _lib


def convolve_same(model, kernel_image):
    """model: (n, H, W) tensor/array; kernel_image: (n or 1, Py, Px).  Returns a device tensor
    (n, H, W) = each plane convolved with its kernel, cropped to the model's shape."""
    torch = _lib.require_gpu()
    as_t = lambda a: a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a))
    m = as_t(model).to(device="cuda", dtype=torch.float32).contiguous()
    k = as_t(kernel_image).to(device="cuda", dtype=torch.float32).contiguous()
    assert m.ndim == 3 and k.ndim == 3
    out = torch.empty_like(m)
    n, H, W = m.shape
    _lib.check(_lib.lib.scarlet_convolve_same(_lib.ptr(m), n, H, W, _lib.ptr(k), k.shape[0], k.shape[1], k.shape[2],
                                              _lib.ptr(out), _lib.stream_ptr()))
    return out
