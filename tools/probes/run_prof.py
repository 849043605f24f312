"""Phase timing of conv_igemm_spx_kernel: s_memtime ticks per K step, waves of the first 64 blocks.
Needs the library built with -DCDF_PROFILE=1 at tools/_ablate/prof/lib_prof.so:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCDF_PROFILE=1 -Iinclude -Icold-diffusion-models_amd/csrc \
        -shared cold-diffusion-models_amd/csrc/*.hip -o tools/_ablate/prof/lib_prof.so"""
import ctypes, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
os.environ["COLDDIFF_LIB"] = os.path.join(REPO, "tools/_ablate/prof/lib_prof.so")
from colddiff import _lib, convdesc as cd
L = _lib.get(); dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ["COLDDIFF_LIB"])
TUNE = _lib.GemmTuning(L).set(halo=0)
P = lambda t: 0 if t is None else t.data_ptr()
S = lambda: torch.cuda.current_stream().cuda_stream
def split(t):
    C = t.shape[-1]; hi = torch.empty(t.shape, dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    L.cdf_split_bf16(P(t), C, P(hi), P(lo), C, t.numel()//C, C, S()); return hi, lo
for (Cin, Cout, H, tile) in [(512, 1024, 16, (128, 128)), (512, 1024, 16, (256, 128)), (64, 128, 128, (256, 128)), (128, 64, 128, (128, 64))]:
    TUNE.set(tile_bm=tile[0], tile_bn=tile[1])
    B, k = 32, 3
    x = torch.randn(B, H, H, Cin, device=dev); y = torch.empty(B, H, H, Cout, device=dev)
    ldk = (Cin + 31) // 32 * 32
    hi = torch.zeros(k*k, Cout, ldk, dtype=torch.int16, device=dev); lo = torch.zeros_like(hi)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    L.cdf_pack_weight_bf16(P(w), P(hi), P(lo), k*k, Cout, Cin, ldk, 1, Cin*k*k, k*k, S())
    xs = split(x); zero = torch.zeros(64, device=dev)
    p = cd.conv_fwd(H, H, k, k, 1, 1, 1, 1, 1)
    for _ in range(3):
        L.cdf_conv_gemm_bf16x(P(xs[0]), P(xs[1]), Cin, P(zero), P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc,
                              0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, TUNE.ptr, S())
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 8 * 6))()
    raw.cdf_debug_read_prof(buf)
    nw = 8 if tile[0] == 256 else 4
    rows = [[buf[(b * nw + wv) * 6 + i] for i in range(6)] for b in range(64) for wv in range(nw)]
    rows = [r for r in rows if r[5] > 0]
    n = rows[0][5]
    avg = [sum(r[i] for r in rows) / len(rows) / n for i in range(4)]
    tot = sum(r[4] for r in rows) / len(rows) / n
    print(f"{Cin}->{Cout}@{H} tile {tile}: K steps {n}; ticks per step: dma-issue {avg[0]:.0f}  reads+mfma {avg[1]:.0f}  vmcnt-wait {avg[2]:.0f}  barrier {avg[3]:.0f}  | loop total {tot:.0f}", flush=True)
