"""Probe: MIOpen fused conv+bias+relu (aten.miopen_convolution_relu / _add_relu) against
conv + rmem_bias_act_nchw for the encoder's layer shapes."""
import sys, time, torch
sys.path.insert(0, '.')
from rmem_amd import hip
dev = 'cuda:0'
shapes = [  # cin, cout, k, stride, H, W
    (64, 64, 1, 1, 121, 213), (64, 64, 3, 1, 121, 213), (64, 256, 1, 1, 121, 213), (256, 64, 1, 1, 121, 213),
    (256, 128, 1, 1, 121, 213), (128, 128, 3, 2, 121, 213), (128, 512, 1, 1, 61, 107), (512, 128, 1, 1, 61, 107),
    (128, 128, 3, 1, 61, 107), (512, 256, 1, 1, 61, 107), (256, 256, 3, 2, 61, 107), (256, 1024, 1, 1, 31, 54),
    (1024, 256, 1, 1, 31, 54), (256, 256, 3, 1, 31, 54)]
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    for cin, cout, k, s, H, W in shapes:
        x = torch.randn(1, cin, H, W, device=dev); w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev); p = k // 2
        def base():
            y = torch.nn.functional.conv2d(x, w, None, s, p)
            return hip.bias_act_nchw_(y, b, None, True)
        def fused():
            return torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [p, p], [1, 1], 1)
        try:
            d = float((base() - fused()).abs().max())
            print(f"{cin:5d}->{cout:5d} k{k} s{s} {H}x{W}: conv+epilogue {t(base):7.1f} us   miopen fused {t(fused):7.1f} us   maxdiff {d:.2e}")
        except Exception as e:
            print("fail", cin, cout, k, repr(e)[:150])
