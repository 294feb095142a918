"""A/B of the training bench with the op-by-op C2f graph (GPU box): python tools/bench_train_ab.py [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_yolo_amd.nn.modules import C2f

C2f.fuse_block_train = False
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
