"""Import alias: the package directory is named after the reference
(`multimodal-image-similarity-search_amd/`), which is not a valid Python identifier, so this module
loads it under the importable name ``mmiss_amd`` (submodules: ``mmiss_amd.utils``, ``mmiss_amd.search``, ...).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "multimodal-image-similarity-search_amd")
_spec = _ilu.spec_from_file_location("mmiss_amd", _os.path.join(_PKG_DIR, "__init__.py"),
                                     submodule_search_locations=[_PKG_DIR])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["mmiss_amd"] = _mod
_spec.loader.exec_module(_mod)
