"""Import shim: the package directory is ``hexl-fpga_amd/`` (hyphenated like the reference's
project name), which Python cannot import by name. ``import hexl_fpga_amd`` loads it."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "hexl-fpga_amd"
_spec = importlib.util.spec_from_file_location(
    "hexl_fpga_amd", _pkg_dir / "__init__.py", submodule_search_locations=[str(_pkg_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hexl_fpga_amd"] = _mod
_spec.loader.exec_module(_mod)
