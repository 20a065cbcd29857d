"""MI355X-native eaQHM analysis hot path (drop-in for the reference's functions.py entry points).

The package directory is named `eaqhm-analysis-and-synthesis-in-python_amd` (not importable by that
name); `import eaqhm_amd` at the repository root loads it under the module name `eaqhm_amd`.
"""
from .functions import (eaQHMAnalysisAndSynthesis, eaQHMAnalysisAndSynthesisBatch, eaqhmLS_complexamps,  # noqa: F401
                        iqhmLS_complexamps, phase_integr_interpolation)
from .hip import HipUnavailable, load_library  # noqa: F401
from .structs import Deterministic, Frame  # noqa: F401
