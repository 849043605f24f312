"""Per-tile phase times of conv_igemm_rowhalo_kernel (prologue / K loop / epilogue incl. store acknowledgement), 100 MHz ticks.
Needs the -DCDF_PROFILE=1 library at tools/_ablate/prof/lib_prof.so (see run_prof.py)."""
import ctypes, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
os.environ["COLDDIFF_LIB"] = os.path.join(REPO, "tools/_ablate/prof/lib_prof.so")
from colddiff import _lib, convdesc as cd
L = _lib.get(); dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ["COLDDIFF_LIB"])
P = lambda t: 0 if t is None else t.data_ptr()
S = lambda: torch.cuda.current_stream().cuda_stream
def split(t):
    C = t.shape[-1]; hi = torch.empty(t.shape, dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    L.cdf_split_bf16(P(t), C, P(hi), P(lo), C, t.numel()//C, C, S()); return hi, lo
TUNE = _lib.GemmTuning(L).set(halo=64 | 47)          # row-halo kernel wherever it applies
for persist in (0,):
    pass
    for (Cin, Cout, H) in [(64, 128, 128), (128, 64, 128), (128, 256, 64), (512, 1024, 16)]:
        B, k = 32, 3
        x = torch.randn(B, H, H, Cin, device=dev); y = torch.empty(B, H, H, Cout, device=dev)
        ldk = (Cin + 31) // 32 * 32
        hi = torch.zeros(k*k, Cout, ldk, dtype=torch.int16, device=dev); lo = torch.zeros_like(hi)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        L.cdf_pack_weight_bf16(P(w), P(hi), P(lo), k*k, Cout, Cin, ldk, 1, Cin*k*k, k*k, S())
        xs = split(x); zero = torch.zeros(64, device=dev)
        p = cd.conv_fwd(H, H, k, k, 1, 1, 1, 1, 1)
        f = lambda: L.cdf_conv_gemm_bf16x(P(xs[0]), P(xs[1]), Cin, P(zero), P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc,
                                          0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, TUNE.ptr, S())
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 4))()
        raw.cdf_debug_read_prof_tile(buf)
        rows = [[buf[b * 4 + i] for i in range(4)] for b in range(256) if buf[b * 4 + 3] > 0]
        n = sum(r[3] for r in rows)
        us = [sum(r[i] for r in rows) / n / 100.0 for i in range(3)]
        print(f"persist {persist} {Cin}->{Cout}@{H}: kernel {e0.elapsed_time(e1)*1e3:.0f} us; per tile (avg over {len(rows)} blocks, {n} tiles): "
              f"prologue {us[0]:.2f} us  K loop {us[1]:.2f} us  epilogue+store ack {us[2]:.2f} us", flush=True)
        b6 = (ctypes.c_ulonglong * (64 * 8 * 6))()
        raw.cdf_debug_read_prof(b6)
        for grp, name in ((range(0, 4), "waves 0-3 (early)"), (range(4, 8), "waves 4-7 (late)")):
            rw = [[b6[(b * 8 + wv) * 6 + i] for i in range(6)] for b in range(64) for wv in grp]
            rw = [r for r in rw if r[5] > 0]
            if rw:
                a5 = [sum(r[i] / r[5] for r in rw) / len(rw) for i in range(5)]
                print(f"    {name}: shader clocks per K step: DMA issue {a5[0]:.0f}  fragment reads {a5[1]:.0f}  MFMAs {a5[2]:.0f}  DMA wait {a5[3]:.0f}  barrier {a5[4]:.0f}  (sum {sum(a5):.0f})", flush=True)
