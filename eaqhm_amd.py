"""Import shim: exposes the package that lives in `eaqhm-analysis-and-synthesis-in-python_amd/`
(a directory name Python cannot import directly) as the module `eaqhm_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eaqhm-analysis-and-synthesis-in-python_amd")
_spec = importlib.util.spec_from_file_location("eaqhm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["eaqhm_amd"] = _mod
_spec.loader.exec_module(_mod)

if __name__ == "__main__":  # python eaqhm_amd.py file.wav --gender female
    from eaqhm_amd.cli import main as _main
    raise SystemExit(_main())
