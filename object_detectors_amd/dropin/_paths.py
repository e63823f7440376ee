"""Makes the `object_detectors_amd` package importable when only this directory is on PYTHONPATH
(the reference puts lmi_utils/ and object_detectors/ on PYTHONPATH via lmi_ai.env:9-17 and imports
top-level names; sourcing object_detectors_amd/dropin/lmi_ai.env does the same for this build)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
