"""Reference-compatible import paths (`src.models...`) so the reference's Hydra `_target_` strings resolve to
the MI355X-native modules in gif-synthesis-with-discrete-diffusion_amd/ (SURVEY.md section 8b)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import gsdd_amd  # noqa: E402,F401
