"""Convolution weight-gradient throughput at the VDM-UNet training shapes (32x32 images, batch 128) with the kernel's parts
switched off one at a time.  usage: [B=128] [ABL=0,1,2,4,...] python tools/wgrad_bench.py   (ABL: BSI_WGRAD_ABL flags, one column each)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsi_amd import _native as N  # noqa: E402


def main():
    B = int(os.environ.get("B", "128"))
    H = 32
    dev = torch.device("cuda:0")
    lib = N.lib()
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    shapes = [("128 -> 128", 128, 0, 128), ("256 -> 128", 256, 0, 128), ("128 (+256 skip) -> 128", 128, 256, 128), ("128 -> 384", 128, 0, 384)]
    M = B * H * H
    for name, Cin, Cin2, Cout in shapes:
        K = 9 * Cin + Cin2
        x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
        x2 = torch.randn(M, max(Cin2, 8), device=dev).to(torch.bfloat16)
        dy = torch.randn(M, Cout, device=dev).to(torch.bfloat16)
        out = torch.empty(Cout, K, device=dev)
        db = torch.empty(Cout, device=dev)
        ws = torch.empty(lib.bsi_conv_wgrad_workspace_bytes(M, Cin, Cin2, Cout, 9), dtype=torch.uint8, device=dev)
        line = f"{name:24s}"
        for abl in os.environ.get("ABL", "0").split(","):
            os.environ["BSI_WGRAD_ABL"] = abl

            def run():
                N.check(lib.bsi_conv_wgrad_bias_nhwc_bf16(N.ptr(dy), Cout, N.ptr(x), N.ptr(x2) if Cin2 else None, N.ptr(zeros), B, H, H, Cin, Cin2,
                                                          Cout, 9, N.ptr(out), N.ptr(db), 0, N.ptr(ws), N.stream()))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f"  [{abl}] {ms * 1e3:6.1f} us {2.0 * M * K * Cout / ms / 1e9:6.1f} TF"
        os.environ["BSI_WGRAD_ABL"] = "0"
        print(line, flush=True)


if __name__ == "__main__":
    main()
