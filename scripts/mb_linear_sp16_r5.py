#!/usr/bin/env python3
"""Round 5: the tile forms of the split-fp16 projection kernel (cfg 0-3: 128 x 128, 80 x 128, 64 x 64, 160 x 128 on
v_mfma_f32_16x16x32_f16 with LDS-DMA operands; cfg 4: the register-staged 64 x 64 kernel), interleaved rounds in one process
(cdna_hip_programming.md rule 24) — profiles/r05_mb_linear_sp16_{dbg,epi,forms,small}.txt were taken with this script at earlier
commits of the round, when the 32x32x16 LDS-DMA forms (since removed) were still in the library — with
 * the error of each form against the fp64 product,
 * the timing-only builds (no DMA / MFMAs only) of each,
 * the IN-KERNEL clock of each form: delta s_memtime / delta s_memrealtime x 100 MHz around the K loop of every workgroup after
   >= 2 s of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS give-back item 6), and from it the cycles per MFMA
   the loop actually takes (loop cycles x SIMD share / MFMAs per wave).
One line per shape and form."""
import ctypes as C
import os, sys, time
from pathlib import Path
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6292
shapes = [(rows, 768, 2304, "qkv"), (rows, 768, 3072, "fc1"), (rows, 768, 768, "out"), (rows, 3072, 768, "fc2"),
          (36335, 768, 2304, "qkv-36k"), (36335, 768, 3072, "fc1-36k"), (36335, 768, 768, "out-36k"), (36335, 3072, 768, "fc2-36k"),
          (10500, 768, 2304, "qkv-10k"), (10500, 768, 3072, "fc1-10k"), (10500, 768, 768, "out-10k"), (10500, 3072, 768, "fc2-10k"),
          (rows, 5120, 1280, "fc2-bigG"), (rows, 1280, 1280, "out-bigG"),
          (640, 768, 2304, "qkv-n100"), (640, 768, 3072, "fc1-n100"), (640, 768, 768, "out-n100"), (640, 3072, 768, "fc2-n100"),
          (1000, 3072, 768, "fc2-keys"), (3072, 768, 768, "out-query"), (3072, 768, 3072, "fc1-query"), (3072, 3072, 768, "fc2-query"),
          (1000, 5120, 1280, "fc2-keys-bigG"), (2100, 768, 2304, "qkv-2k"), (2100, 3072, 768, "fc2-2k"),
          (rows, 1280, 3840, "qkv-bigG"), (rows, 1280, 5120, "fc1-bigG")]
# whole rounds of 128 x 128 tiles on 512 workgroup slots (tile-quantisation check): 504 / 1008 / 1512 tiles
shapes += [(3584, 768, 2304, "qkv-504t"), (7168, 768, 2304, "qkv-1008t"), (2688, 768, 3072, "fc1-504t"), (5376, 768, 3072, "fc1-1008t"),
           (8064, 768, 3072, "fc1-1512t")]
if os.environ.get("MB_SHAPES"):
    shapes = [sh for sh in shapes if sh[3] in os.environ["MB_SHAPES"].split(",")]
lib = hip.load()
lib.emcid_debug_linear_sp16_stamps.restype = C.c_int
lib.emcid_debug_linear_sp16_stamps.argtypes = [C.c_void_p]

# (cfg, name, tile rows, tile cols, waves per workgroup, MFMAs per wave and 32-deep stage, nominal cycles per MFMA)
FORMS = [(0, "128x128", 128, 128, 4, 48, 16), (1, "80x128", 80, 128, 4, 30, 16), (2, "64x64", 64, 64, 4, 12, 16),
         (3, "160x128", 160, 128, 4, 60, 16)]
if os.environ.get("MB_FORMS"):
    FORMS = [f for f in FORMS if str(f[0]) in os.environ["MB_FORMS"].split(",")]


def rounds(fns, n_rounds=7, n=20):
    """interleaved timing: every form once per round; returns the per-form list of per-round means (us)"""
    for f in fns:
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    out = [[] for _ in fns]
    for _ in range(n_rounds):
        for i, f in enumerate(fns):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            out[i].append((time.perf_counter() - t0) / n * 1e6)
    return out


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


for M, K, N, name in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    y = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    ref = F.linear(x.double(), w.double(), b.double())
    top = ref.abs().max().item()
    xs, ws = hip.split_rows(x), hip.split_rows(w)
    yauto = hip.linear_sp(xs, ws, b)
    print(f"== {name}: {M} x {K} -> {N}   ({fl * 3 / 1e9:.1f} GF of f16 MFMA work)", flush=True)
    errs = {}
    for cfg, cn, *_ in FORMS:
        yy = hip.linear_sp(xs, ws, b, cfg=cfg)
        e = (yy.double() - ref).abs()
        errs[cfg] = (e.max().item() / top, e.pow(2).mean().sqrt().item() / top, (yy - yauto).abs().max().item() / top)
    fns = [(lambda c=cfg: hip.linear_sp(xs, ws, b, out=y, cfg=c)) for cfg, *_ in FORMS]
    fns.append(lambda: hip.linear_sp(xs, ws, b, out=y, cfg=4))
    fns.append(lambda: hip.linear_sp(xs, ws, b, out=y, cfg=-1))
    extra_cfgs = [int(v) for v in os.environ.get("MB_EXTRA_CFGS", "").split(",") if v]
    for c in extra_cfgs:
        fns.append(lambda c=c: hip.linear_sp(xs, ws, b, out=y, cfg=c))
    tt = rounds(fns)
    n_extra = len(extra_cfgs)
    t_auto = med(tt[-1 - n_extra])
    print(f"   auto                      {t_auto:7.1f} us  {fl / t_auto / 1e6:6.1f} TF-equivalent | register-staged 64x64 {med(tt[-2 - n_extra]):7.1f} us"
          + "".join(f" | cfg {c}: {med(tt[len(tt) - n_extra + i]):7.1f} us (err {((hip.linear_sp(xs, ws, b, cfg=c).double() - ref).abs().max().item() / top):.1e})" for i, c in enumerate(extra_cfgs)), flush=True)
    dbg = {}
    if os.environ.get("MB_DBG", "1") == "1":
        dfns, keys = [], []
        for cfg, *_ in FORMS:
            for d, dn in ((16, "no-dma"), (48, "mfma-only")):
                if os.environ.get("MB_DBG_FORMS", "1") != "1":
                    continue
                if cfg != 0:
                    continue            # timing-only builds exist for the 128 x 128 form
                dfns.append(lambda c=cfg + d: hip.linear_sp(xs, ws, b, out=y, cfg=c))
                keys.append((cfg, dn))
        for k, v in zip(keys, rounds(dfns, n_rounds=3)):
            dbg[k] = med(v)
    for (cfg, cn, bm, bn, waves, mfma_per_stage, nominal), t in zip(FORMS, tt):
        tm = med(t)
        if os.environ.get("MB_RB") or os.environ.get("MB_STAGGER"):
            # the tile order's super-row height (row tiles) and the first round's stagger (mode:sleeps of 2 048 cycles),
            # interleaved rounds; "0" / "1:0" = what the library does.  Needs the diagnostic entry of
            # profiles/r05_linear_stagger_dropped.patch (measured: no gain from either, profiles/r05_mb_linear_sp16_order.txt)
            knobs = [(int(v), 0, 1) for v in os.environ.get("MB_RB", "").split(",") if v]
            knobs += [(0, int(v.split(":")[1]), int(v.split(":")[0])) for v in os.environ.get("MB_STAGGER", "").split(",") if v]
            def with_knob(kn, c=cfg):
                lib.emcid_debug_linear_sp16_order(*kn)
                hip.linear_sp(xs, ws, b, out=y, cfg=c)
            res = rounds([(lambda kn=kn: with_knob(kn)) for kn in knobs], n_rounds=5)
            lib.emcid_debug_linear_sp16_order(0, -1, 1)
            print(f"      {cn}: (super-row, stagger, mode) -> us: " + " ".join(f"{kn}:{med(v):.1f}" for kn, v in zip(knobs, res)), flush=True)
        if cfg != 0:           # the stamped build exists for the 128 x 128 form
            e = errs[cfg]
            print(f"   {cn:24s} {tm:7.1f} us  {fl / tm / 1e6:6.1f} TF-eq = {3 * fl / tm / 1e9:6.3f} PF executed (min {min(t):6.1f}) | err max {e[0]:.1e} rms {e[1]:.1e} vs auto {e[2]:.1e}", flush=True)
            continue
        # in-kernel clock: >= 2 s of back-to-back launches, then the stamps of the last one
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        nwg = ((tiles + 7) // 8) * 8
        stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
        lib.emcid_debug_linear_sp16_stamps(C.c_void_p(stamps.data_ptr()))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            for _ in range(50):
                hip.linear_sp(xs, ws, b, out=y, cfg=cfg)
            torch.cuda.synchronize()
        stamps.zero_()
        hip.linear_sp(xs, ws, b, out=y, cfg=cfg)
        torch.cuda.synchronize()
        lib.emcid_debug_linear_sp16_stamps(None)
        st = stamps.view(-1, 8).cpu()
        st = st[st[:, 3] > st[:, 2]]
        # a workgroup's life in microseconds (100 MHz ticks): entry -> loop start, loop, loop end -> exit; and the launch's span
        pro = ((st[:, 2] - st[:, 4]).double() * 0.01).median().item()
        lp = ((st[:, 3] - st[:, 2]).double() * 0.01).median().item()
        epi = ((st[:, 5] - st[:, 3]).double() * 0.01).median().item()
        span = (st[:, 5].max() - st[:, 4].min()).item() * 0.01
        last_start = (st[:, 4].max() - st[:, 4].min()).item() * 0.01
        clk = ((st[:, 1] - st[:, 0]).double() / (st[:, 3] - st[:, 2]).double() * 0.1)      # GHz
        loop_cycles = (st[:, 1] - st[:, 0]).double().median().item()
        stages = K // 32
        # a SIMD carries waves / 4 waves of the workgroup (and as many of the other resident workgroup for the 4-wave tiles:
        # two workgroups per compute unit); cycles per MFMA if this workgroup had the SIMD to itself:
        per_mfma = loop_cycles / (stages * mfma_per_stage * (waves / 4))
        e = errs[cfg]
        extra = "".join(f" | {dn} {dbg[(cfg, dn)]:6.1f} us" for dn in ("no-dma", "mfma-only") if (cfg, dn) in dbg)
        print(f"   {cn:24s} {tm:7.1f} us  {fl / tm / 1e6:6.1f} TF-eq = {3 * fl / tm / 1e9:6.3f} PF executed (min {min(t):6.1f}) | err max {e[0]:.1e} rms {e[1]:.1e}"
              f" vs auto {e[2]:.1e} | clock {clk.median().item():.3f} GHz (p10 {clk.quantile(0.1).item():.3f}) loop {loop_cycles:8.0f} cyc ="
              f" {per_mfma:5.1f} cyc/MFMA at the workgroup's SIMD share (nominal {nominal}) | workgroup: prologue {pro:5.1f} + loop {lp:5.1f} + epilogue {epi:5.1f} us,"
              f" launch span {span:6.1f} us, last workgroup starts at {last_start:5.1f}{extra}", flush=True)
        if os.environ.get("MB_TIMELINE"):
            # workgroups alive (and inside their K loop) every 4 us of the launch, and a workgroup's lifetime by start order
            t_in = (st[:, 4] - st[:, 4].min()).double() * 0.01
            t_l0 = (st[:, 2] - st[:, 4].min()).double() * 0.01
            t_l1 = (st[:, 3] - st[:, 4].min()).double() * 0.01
            t_out = (st[:, 5] - st[:, 4].min()).double() * 0.01
            line = []
            for t in range(0, int(span) + 4, 4):
                line.append(f"{t}:{int(((t_in <= t) & (t_out > t)).sum())}/{int(((t_l0 <= t) & (t_l1 > t)).sum())}")
            print("      alive/in-loop at t us: " + " ".join(line))
            order = torch.argsort(t_in)
            life = (t_out - t_in)[order]
            q = len(life) // 8
            print("      lifetime (us) by start order, eighths: " + " ".join(f"{life[i * q:(i + 1) * q].mean().item():.1f}" for i in range(8))
                  + f" | starts (us) eighth means: " + " ".join(f"{t_in[order][i * q:(i + 1) * q].mean().item():.1f}" for i in range(8)), flush=True)
