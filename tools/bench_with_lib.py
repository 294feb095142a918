"""Run bench.py against another build of libdyolo.so (GPU box): python tools/bench_with_lib.py <lib.so> [bench.py args ...].
Only a -DDYOLO_ABLATE build (make ABLATE=1 OUT=...) reads the DYOLO_* probes, so A/B runs of kernel variants go through here."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_yolo_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
