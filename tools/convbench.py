import os
"""Time cdf_conv_gemm / cdf_conv_wgrad at the CelebA-128 layer shapes (B from env KB_B, default 32)."""
import json, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import _lib, convdesc as cd
L = _lib.get(); dev = torch.device("cuda:0")
# tuning: the explicit cdf_gemm_tuning argument (the library has no setters); KB_* environment knobs fill its fields
TUNE = _lib.GemmTuning(L)
if os.environ.get('KB_TILE'):
    bm_, bn_ = (int(v) for v in os.environ['KB_TILE'].split('x'))
    TUNE.set(tile_bm=bm_, tile_bn=bn_)
if os.environ.get('KB_HALO'):
    h_ = [int(v) for v in os.environ['KB_HALO'].split(',')]
    TUNE.set(halo=h_[0], halo_min_tiles=h_[1] if len(h_) > 1 else 1)
for var_, field_ in (('KB_ROW3', 'wgrad_row3'), ('KB_WSWZ', 'wgrad_swizzle'), ('KB_HALO_BM', 'halo_bm'), ('KB_DEPHASE', 'dephase'),
                     ('KB_SMALL_N64', 'small_n64'), ('KB_MAX_BM', 'max_bm'), ('KB_ROWHALO_STREAM', 'rowhalo_stream')):
    if os.environ.get(var_):
        TUNE.set(**{field_: int(os.environ[var_])})
SINGLE = os.environ.get('KB_SINGLE', '0') == '1'      # hi-only planes: single-pass bf16 (NS = 1 kernels)

S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: 0 if t is None else t.data_ptr()
r4 = lambda c: (c + 3) // 4 * 4
B = int(os.environ.get("KB_B", "32")); iters = int(os.environ.get("KB_ITERS", "5"))
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
shapes = [(64,128,128,3),(128,64,128,3),(128,256,64,3),(256,128,64,3),(256,512,32,3),(512,256,32,3),(512,1024,16,3),(1024,512,16,3),(1024,2048,16,3),(64,384,128,1)]
only = os.environ.get("KB_ONLY")
if os.environ.get("KB_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("-")) for t in os.environ["KB_SHAPES"].split(",")]
for (Cin,Cout,H,k) in shapes:
    if os.environ.get("KB_F32", "1") != "1": break
    if only and only != f"{Cin}-{Cout}-{H}": continue
    x = torch.randn(B,H,H,r4(Cin),device=dev); w = torch.randn(k*k,Cin,r4(Cout),device=dev)*0.05; y = torch.empty(B,H,H,r4(Cout),device=dev)
    p = cd.conv_fwd(H,H,k,k,1,k//2,k//2,k//2,k//2)
    ms = timeit(lambda: L.cdf_conv_gemm(P(x),x.shape[-1],P(w),w.shape[-1],P(y),y.shape[-1],B,H,H,Cin,H,H,Cout,H,H,1,1,1,p.desc,0,0,0,0,0,0,0,0,0,0,0,0,0,1,0,0,0,1,0,0,0,S()))
    fl = 2.0*B*H*H*Cin*Cout*k*k
    print(f"fwd   {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF", flush=True)
    if os.environ.get("KB_WGRAD","1") == "1":
        wg = cd.conv_wgrad(H,H,k,k,1,k//2,k//2,k//2,k//2); M=B*H*H
        ns = L.cdf_wgrad_nsplit(M,Cin,Cout,k*k); ws = torch.empty(ns,k*k,Cin,r4(Cout),device=dev)
        ms = timeit(lambda: L.cdf_conv_wgrad(P(x),x.shape[-1],P(y),y.shape[-1],P(ws),r4(Cout),B,H,H,H,H,1,H,H,1,Cin,Cout,k*k,wg.desc,ns,1,0,0,0,0,S()))
        print(f"wgrad {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF (ns={ns})", flush=True)
# ---- split-precision bf16 MFMA variants --------------------------------------------------------------
if os.environ.get("KB_SP", "1") == "1":
    for (Cin,Cout,H,k) in shapes:
        if only and only != f"{Cin}-{Cout}-{H}": continue
        x = torch.randn(B,H,H,r4(Cin),device=dev); y = torch.empty(B,H,H,r4(Cout),device=dev)
        ldk = (Cin+31)//32*32
        hi = torch.zeros(k*k,Cout,ldk,dtype=torch.int16,device=dev); lo = torch.zeros_like(hi)
        w = torch.randn(Cout,Cin,k,k,device=dev)*0.05
        L.cdf_pack_weight_bf16(P(w),P(hi),P(lo),k*k,Cout,Cin,ldk,1,Cin*k*k,k*k,S())
        p = cd.conv_fwd(H,H,k,k,1,k//2,k//2,k//2,k//2)
        fl = 2.0*B*H*H*Cin*Cout*k*k
        for split in (3,1):
            ms = timeit(lambda: L.cdf_conv_gemm_bf16(P(x),x.shape[-1],P(hi),P(lo),ldk,P(y),y.shape[-1],B,H,H,Cin,H,H,Cout,H,H,1,1,1,p.desc,0,0,0,0,0,0,0,0,0,0,0,0,split,S()))
            print(f"sp{split}   {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF-equiv", flush=True)
if os.environ.get("KB_SPW", "1") == "1":
    for (Cin,Cout,H,k) in shapes:
        if only and only != f"{Cin}-{Cout}-{H}": continue
        if Cin < 64 or Cout < 64: continue
        x = torch.randn(B,H,H,r4(Cin),device=dev); y = torch.randn(B,H,H,r4(Cout),device=dev)
        wg = cd.conv_wgrad(H,H,k,k,1,k//2,k//2,k//2,k//2); M=B*H*H
        tiles = ((Cin+127)//128)*((Cout+127)//128)*k*k
        ns = max(1, min(512//tiles if tiles <= 512 else 1, M//512))
        ws = torch.empty(ns,k*k,Cin,r4(Cout),device=dev)
        ms = timeit(lambda: L.cdf_conv_wgrad_bf16(P(x),x.shape[-1],P(y),y.shape[-1],P(ws),r4(Cout),B,H,H,H,H,1,H,H,1,Cin,Cout,k*k,wg.desc,ns,0,S()))
        fl = 2.0*B*H*H*Cin*Cout*k*k
        print(f"spW   {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF-equiv (ns={ns})", flush=True)
# ---- pre-split operand kernels ---------------------------------------------------------------------------
if os.environ.get("KB_SPX", "1") == "1":
    zero = torch.zeros(64, device=dev)
    def split(t):
        C = t.shape[-1]; hi = torch.empty(t.shape, dtype=torch.int16, device=dev); lo = None if SINGLE else torch.empty_like(hi)
        L.cdf_split_bf16(P(t), C, P(hi), P(lo), C, t.numel()//C, C, S()); return hi, lo
    for (Cin,Cout,H,k) in shapes:
        if only and only != f"{Cin}-{Cout}-{H}": continue
        if Cin % 8 or Cout % 8: continue
        x = torch.randn(B,H,H,Cin,device=dev); y = torch.empty(B,H,H,r4(Cout),device=dev); gy = torch.randn(B,H,H,Cout,device=dev)
        ldk = (Cin+31)//32*32
        hi = torch.zeros(k*k,Cout,ldk,dtype=torch.int16,device=dev); lo = None if SINGLE else torch.zeros_like(hi)
        w = torch.randn(Cout,Cin,k,k,device=dev)*0.05
        L.cdf_pack_weight_bf16(P(w),P(hi),P(lo),k*k,Cout,Cin,ldk,1,Cin*k*k,k*k,S())
        ms = timeit(lambda: split(x)); print(f"split {Cin:5d} ch @{H:3d}: {ms:8.3f} ms {8.0*x.numel()/ms/1e6:7.1f} GB/s", flush=True)
        xs = split(x); gs = split(gy)
        p = cd.conv_fwd(H,H,k,k,1,k//2,k//2,k//2,k//2)
        fl = 2.0*B*H*H*Cin*Cout*k*k
        ms = timeit(lambda: L.cdf_conv_gemm_bf16x(P(xs[0]),P(xs[1]),Cin,P(zero),P(hi),P(lo),ldk,P(y),y.shape[-1],B,H,H,Cin,H,H,Cout,H,H,1,1,1,p.desc,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,TUNE.ptr,S()))
        print(f"spx   {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF-equiv", flush=True)
        if os.environ.get("KB_EPI", "1") == "1":      # the ConvNeXt conv1 form: bias + GELU, pre-activation kept, output as bf16 planes only
            bias = torch.randn(Cout, device=dev); pre = torch.empty(B,H,H,Cout,device=dev)
            yh = torch.empty(B,H,H,Cout,dtype=torch.int16,device=dev); yl = None if SINGLE else torch.empty_like(yh)
            ms = timeit(lambda: L.cdf_conv_gemm_bf16x(P(xs[0]),P(xs[1]),Cin,P(zero),P(hi),P(lo),ldk,0,Cout,B,H,H,Cin,H,H,Cout,H,H,1,1,1,p.desc,P(bias),0,0,0,0,P(pre),Cout,0,0,1,0,0,P(yh),P(yl),Cout,0,0,TUNE.ptr,S()))
            print(f"spxG  {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF-equiv (bias+GELU, pre + planes out)", flush=True)
        wg = cd.conv_wgrad(H,H,k,k,1,k//2,k//2,k//2,k//2); M=B*H*H
        row3 = L.cdf_conv_wgrad_bf16x_is_row3(H, H, Cin, Cout, k*k, 1 if k == 3 else 0, TUNE.ptr)
        tiles = ((Cin+127)//128)*((Cout+127)//128)*(3 if row3 else k*k)
        slots = 256 if row3 else 512
        best, bc = 1, None
        for ns_ in range(1, min(M//512, 256)+1):
            c_ = -(-tiles*ns_//slots)/ns_
            if bc is None or c_ < bc - 1e-9: best, bc = ns_, c_
        ns = best
        ws = torch.empty(ns,k*k,Cin,r4(Cout),device=dev)
        ms = timeit(lambda: L.cdf_conv_wgrad_bf16x(P(xs[0]),P(xs[1]),Cin,P(gs[0]),P(gs[1]),Cout,P(zero),P(ws),r4(Cout),B,H,H,H,H,1,H,H,1,Cin,Cout,k*k,wg.desc,ns,0,TUNE.ptr,S()))
        print(f"spxW  {Cin:5d}->{Cout:5d} @{H:3d} k{k}: {ms:8.3f} ms {fl/ms/1e9:7.1f} TF-equiv (ns={ns})", flush=True)
