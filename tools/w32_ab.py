"""A/B of the MFMA shape inside a production kernel (GPU box, ablate build): the flat-K 128 x 128 tile of conv_gemm_fk.hip on
v_mfma_f32_16x16x32 (product) against the same tile, loop and epilogue on v_mfma_f32_32x32x16 (DYOLO_FK_W32=1).
usage: python tools/w32_ab.py [--dtype fp16] [--batch 8] [shape ...]   shape = cin,cout,k,s,H   (make ABLATE=1 OUT=../lib_ablate first)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DYOLO_FK_BN"] = "128"
os.environ["DYOLO_NO_VGEMM"] = os.environ["DYOLO_NO_GLDS"] = "1"  # the flat-K kernel takes every shape
import torch
from drone_yolo_amd import _lib
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("shapes", nargs="*", default=["128,128,3,1,160", "256,256,3,1,80", "512,512,3,1,40", "128,256,3,2,160", "256,512,3,2,80",
                                               "384,128,1,1,160", "768,256,1,1,80", "1024,512,1,1,40", "256,256,1,1,80"])
a = ap.parse_args()
_lib.LIB_PATH = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "drone-yolo_amd", "lib_ablate", "libdyolo.so"))
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
dev = torch.device("cuda", 0)


def timed(x, pc, y):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.iters):
        H.conv2d(x, pc, out=y)
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / a.iters * 1e3


for sh in a.shapes:
    cin, cout, k, s, hh = (int(v) for v in sh.split(","))
    x = torch.randn(a.batch, hh, hh, cin, device=dev).to(dt).permute(0, 3, 1, 2)
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    pc_bias = torch.randn(cout) * 0.1
    pc = H.PackedConv(w, pc_bias, s, k // 2, 1, True, dt, dev, halo=False)
    os.environ["DYOLO_FK_W32"] = "0"
    y0 = H.conv2d(x, pc)
    n0 = H.last_kernel_name()
    os.environ["DYOLO_FK_W32"] = "1"
    y1 = H.conv2d(x, pc)
    n1 = H.last_kernel_name()
    torch.cuda.synchronize()
    diff = (y0.float() - y1.float()).abs().max().item()
    ref = torch.nn.functional.silu(torch.nn.functional.conv2d(x.float(), w.to(dev).to(dt).float(), pc_bias.to(dev), s, k // 2))
    e0, e1 = (y0.float() - ref).abs().max().item(), (y1.float() - ref).abs().max().item()
    ndiff = int((y0 != y1).sum().item())
    ulp = (y0.float().abs().max().item()) * (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7)
    t = {0: [], 1: []}
    for _ in range(a.reps):
        for v in (0, 1):
            os.environ["DYOLO_FK_W32"] = str(v)
            t[v].append(timed(x, pc, y0 if v == 0 else y1))
    fl = 2.0 * a.batch * y0.shape[2] * y0.shape[3] * cout * cin * k * k
    m0, m1 = min(t[0]), min(t[1])
    print(f"{sh:<18s} B={a.batch} {a.dtype} {n0} vs {n1}: max|diff| {diff:.3g} in {ndiff} of {y0.numel()} outputs (one output ulp at max {ulp:.3g}; max error against the fp32 convolution {e0:.3g} / {e1:.3g})  16x16x32 {m0:8.1f} us {fl / m0 / 1e6:6.1f} TF | "
          f"32x32x16 {m1:8.1f} us {fl / m1 / 1e6:6.1f} TF | ratio {m0 / m1:.3f}  runs {[round(v, 1) for v in t[0]]} {[round(v, 1) for v in t[1]]}", flush=True)
