"""Import alias: the package directory is named `instance-segmentation-attention_amd` (not a valid
Python identifier); this module loads it under the name `isa_amd` so `import isa_amd.engine` works."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "instance-segmentation-attention_amd")
_spec = importlib.util.spec_from_file_location("isa_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["isa_amd"] = _mod
_spec.loader.exec_module(_mod)
