"""Implicit-GEMM convolution throughput at the VDM-UNet shapes (32x32 images, 128 output channels).
usage: [B=128] [ABL=0,1,2,...] python tools/conv_bench.py   (ABL: flags of bsi_conv_set_ablation, one column each; bits 1..64 switch kernel
parts off and need a laboratory build: make -C bsi_amd/csrc LAB=1 OUTDIR=/tmp/lab && BSI_HIP_LIB=/tmp/lab/libbsi_hip.so ...)"""
import ctypes as C
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
    # (name, Cin, Cin2, Cout, taps, epilogue)
    shapes = [("conv1 128->128 bf16", 128, 0, 128, 9, N.CONV_BIAS_BF16), ("conv2 128->128 f32+resid", 128, 0, 128, 9, N.CONV_BIAS_RESID_F32),
              ("up conv1 256->128 bf16", 256, 0, 128, 9, N.CONV_BIAS_BF16), ("up conv2 128(+256 skip)->128 f32", 128, 256, 128, 9, N.CONV_BIAS_RESID_F32),
              ("qkv 128->384 bf16", 128, 0, 384, 9, N.CONV_BIAS_BF16), ("dgrad 384->128 bf16", 384, 0, 128, 9, N.CONV_BIAS_BF16)]
    M = B * H * H
    for name, Cin, Cin2, Cout, taps, epi in shapes:
        K = taps * Cin + Cin2
        x = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
        x2 = torch.randn(M, max(Cin2, 1), device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Cout, device=dev)
        f32 = epi == N.CONV_BIAS_RESID_F32
        out = torch.empty(M, Cout, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        res = torch.randn(M, Cout, device=dev) if f32 else None
        a = N.ConvArgs(x=x.data_ptr(), w=w.data_ptr(), bias=bias.data_ptr(), zeros=zeros.data_ptr(), B=B, H=H, W=H, Cin=Cin, Cin2=Cin2,
                       Cout=Cout, taps=taps, ldo=Cout, out=out.data_ptr(), epilogue=epi)
        if Cin2:
            a.x2 = x2.data_ptr()
        if res is not None and not Cin2:
            a.resid = res.data_ptr()
        line = f"{name:36s}"
        for abl in [int(v) for v in os.environ.get("ABL", "0").split(",")]:
            N.check(lib.bsi_conv_set_ablation(abl))
            for _ in range(3):
                N.check(lib.bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                N.check(lib.bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f"  [{abl}] {ms * 1e3:6.1f} us {2.0 * M * K * Cout / ms / 1e9:6.1f} TF"
        N.check(lib.bsi_conv_set_ablation(0))
        print(line, flush=True)


if __name__ == "__main__":
    main()
