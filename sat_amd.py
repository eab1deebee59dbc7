"""Import alias: the product package directory is named after the reference repo
(`show-attend-and-tell-pytorch-lightning_amd/`), which is not a Python identifier.
``import sat_amd`` loads that directory as the package ``sat_amd``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "show-attend-and-tell-pytorch-lightning_amd")
_spec = importlib.util.spec_from_file_location("sat_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["sat_amd"] = _pkg
_spec.loader.exec_module(_pkg)
