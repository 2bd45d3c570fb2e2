"""The one door between far_amd and torch / vendor-library executions of its operators.

The product has ONE execution path per operator: the HIP kernels of libfar_hip.so (plus, in GPU training only, autograd through
`nn.Module` calls of small parameter containers and the two compositions of far_amd/train_glue.py that have no backward kernel).
Everything else -- CPU tensors, the `hip_training = False` comparison legs, `materialize_conf` in training (the reference's dense
differentiable confidence matrix) -- is test and benchmark infrastructure.  It lives OUTSIDE the package (tests/vendor_ops.py) and is
reachable only after someone installed it here:

    import tests.vendor_ops, far_amd._vendor
    far_amd._vendor.install(tests.vendor_ops)        # tests/conftest.py does this; bench.py --vendor-train; tools/make_goldens.py

Without it those paths raise FarHipError -- there is no silent CPU or eager fallback.
"""
import torch

from ._lib import FarHipError

_impl = None


def install(impl):
    """impl: a module / object with the functions of tests/vendor_ops.py.  None uninstalls."""
    global _impl
    _impl = impl


def installed():
    return _impl is not None


def require(what):
    """Gate of a path that is not part of the product.  Returns the installed helper."""
    if _impl is None:
        raise FarHipError(f'{what}: far_amd has no CPU / eager / vendor-library path for this -- its operators run on the HIP kernels of '
                          'libfar_hip.so (GPU tensors).  The torch compositions used by the CPU tests and the vendor comparison legs live in '
                          'tests/vendor_ops.py; install them with far_amd._vendor.install(tests.vendor_ops).')
    return _impl


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)
