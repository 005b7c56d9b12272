"""Loader for the product package.  Its directory is named `biogpt.cpp_amd` (the layout the task
prescribes), which is not an importable Python identifier, so it is imported by path under the
module name `biogpt_cpp_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "biogpt_cpp_amd" in sys.modules:
        return sys.modules["biogpt_cpp_amd"]
    pkg_dir = os.path.join(_ROOT, "biogpt.cpp_amd")
    spec = importlib.util.spec_from_file_location("biogpt_cpp_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["biogpt_cpp_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
