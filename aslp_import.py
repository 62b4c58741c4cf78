"""Registers the package directory `kaldi-aslp_amd/` (hyphenated like the reference repo
name) as the importable module `kaldi_aslp_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "kaldi_aslp_amd" in sys.modules:
        return sys.modules["kaldi_aslp_amd"]
    pkg_dir = os.path.join(_ROOT, "kaldi-aslp_amd")
    spec = importlib.util.spec_from_file_location(
        "kaldi_aslp_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["kaldi_aslp_amd"] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules["kaldi_aslp_amd"]
        raise
    return mod
