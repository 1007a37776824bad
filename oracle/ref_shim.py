"""Import the reference's own Python modules in the BUILD CONTAINER ONLY.

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box and nothing
under tests/ (gpu), smoke() or bench.py may call this at run time; it is used by
``oracle/gen_golden.py`` to produce the committed fixtures in tests/golden/.

Absent third-party modules (pywt, torchvision, cv2, skimage, tkinter) are
replaced by inert stand-ins exactly as SURVEY.md Appendix A records; the only
pywt entry points the Haar path touches are ``Wavelet('haar').dec_*/rec_*`` and
``dwt_coeff_len`` (transform2d.py:23-25,92-94; lowlevel.py:153).  ``train.py`` is
never imported (script with side effects, remote VGG fetch -- SURVEY fact 5).
"""
import math
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(REFERENCE_ROOT)


def load():
    import torch
    if "pywt" not in sys.modules:
        pywt = types.ModuleType("pywt")

        class Wavelet:
            def __init__(self, name):
                assert name in ("haar", "db1")
                s = 1 / math.sqrt(2)
                self.dec_lo, self.dec_hi, self.rec_lo, self.rec_hi = [s, s], [-s, s], [s, s], [s, -s]
        pywt.Wavelet = Wavelet
        pywt.dwt_coeff_len = lambda N, L, mode: (N + 1) // 2 if mode in ("per", "periodization") else (N + L - 1) // 2
        sys.modules["pywt"] = pywt

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = _Stub(self.__name__ + "." + k)
            setattr(self, k, m)
            return m

        def __call__(self, *a, **k):
            return None
    for n in ["tkinter", "cv2", "torchvision", "torchvision.transforms", "torchvision.models", "skimage", "skimage.metrics"]:
        sys.modules.setdefault(n, _Stub(n))
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self      # utils.py:97,110 call .cuda() on the masks
    for p in (os.path.join(REFERENCE_ROOT, "pytorch_wavelets"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import model as ref_model      # noqa: E402
    import utils as ref_utils      # noqa: E402
    import ssim as ref_ssim        # noqa: E402
    from pytorch_wavelets import DWTForward, DWTInverse   # noqa: E402
    return types.SimpleNamespace(model=ref_model, utils=ref_utils, ssim=ref_ssim, DWTForward=DWTForward, DWTInverse=DWTInverse)
