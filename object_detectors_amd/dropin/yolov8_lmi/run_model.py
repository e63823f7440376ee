"""Drop-in name for the reference's folder runner (object_detectors/yolov8_lmi/run_model.py): same command line,
implemented by object_detectors_amd/run_folder.py (batched on the GPU; no annotated images are drawn)."""
import _paths  # noqa: F401
from object_detectors_amd.run_folder import main

if __name__ == '__main__':
    raise SystemExit(main())
