"""Import shim: the package directory is named ``alice-codec_amd`` (not a valid Python
identifier), so ``import alice_codec_amd`` lands here and loads that directory as the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "alice-codec_amd")
_spec = importlib.util.spec_from_file_location(
    "alice_codec_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["alice_codec_amd"] = _mod
_spec.loader.exec_module(_mod)
