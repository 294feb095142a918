"""Small host-side utilities (logger, defaults) — reference: ultralytics/utils/__init__.py."""
import logging
import os

LOGGER = logging.getLogger("drone_yolo_amd")
if not LOGGER.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(message)s"))
    LOGGER.addHandler(_h)
    LOGGER.setLevel(logging.INFO if os.environ.get("DYOLO_VERBOSE", "1") != "0" else logging.WARNING)
    LOGGER.propagate = False

RANK = int(os.getenv("RANK", -1))
LOCAL_RANK = int(os.getenv("LOCAL_RANK", -1))
