"""kaldi-aslp_amd: MI355X-native hot path of robin1001/kaldi-aslp's aslp-nnet training step.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it
through `aslp_import.py` at the repo root, which registers it as module `kaldi_aslp_amd`.

Layout:
  csrc/  hand-written HIP kernels for gfx950 + the C ABI of include/*.h
  nnet/  C++ host engine mirroring the reference's Component / Nnet / loss API
  _lib.py  ctypes binding of libaslp_hip.so  (fails loudly if the .so is missing)
  ops.py   torch.Tensor convenience wrappers over the C ABI (device pointers only)
"""
from . import _lib  # noqa: F401  (raises if libaslp_hip.so is absent)
from ._lib import lib, MatrixDim, Dim3, check_error  # noqa: F401
from . import ops  # noqa: F401
from . import nnet  # noqa: F401
from .nnet import Nnet, Xent, WarpCtc, Ctc, MatrixRandomizer, randomizer_mask  # noqa: F401
