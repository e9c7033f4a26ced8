"""GPU box, diagnostic library only (gemm_conv.hip built with -DDMX_GEMM_STAMPS, loaded through DMX_LIB_PATH): where a workgroup of the
8-wave implicit-GEMM tiles spends its life on the big layers of the step.  Per layer: launch time, per-workgroup mean microseconds of
ring prologue | K loop | epilogue (and the part of it spent waiting for store acknowledgements), the K-loop rate per CU, the gap between consecutive workgroups on one CU, and how far apart the
workgroups of one launch start their epilogues (all CUs bursting to HBM at once or not)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from diffmusic_amd import _lib as L
import tune_tiles as T
lib = L.lib()
lib.dmx_gemm_stamps_read.argtypes = [C.c_void_p]; lib.dmx_gemm_stamps_read.restype = C.c_int
TILE = {1: (256, 256), 7: (320, 256), 19: (512, 128), 2: (256, 128), 8: (320, 128), 9: (192, 256), 10: (192, 128)}
LAYERS = [  # M, N, K, taps, flags, cfg, what
    (160032, 256, 768, 3, 289, 7, "hifigan C=256 k=3 fwd"), (160032, 256, 768, 3, 805, 7, "hifigan C=256 k=3 bwd"),
    (160032, 256, 2816, 11, 289, 7, "hifigan C=256 k=11 fwd"), (160032, 256, 2816, 11, 805, 7, "hifigan C=256 k=11 bwd"),
    (40008, 512, 1536, 3, 289, 7, "hifigan C=512 k=3 fwd"), (40008, 512, 5632, 11, 805, 7, "hifigan C=512 k=11 bwd"),
    (128000, 256, 2304, 9, 5, 1, "vae C=256 3x3"), (32000, 512, 4608, 9, 5, 1, "vae C=512 3x3"), (128000, 256, 1024, 4, 1, 1, "vae up2x C=256"),
    (512000, 128, 1152, 9, 5, 19, "vae C=128 3x3"),
]
if len(sys.argv) > 1 and sys.argv[1] == "few":
    # the same tiles with only a few workgroups on the chip: what a CU's prologue / epilogue cost when HBM is not shared with 255 others
    LAYERS = [(320 * n, 256, 768, 3, f, 7, f"{n} tiles, flags {f}") for n in (8, 32, 64, 128, 256) for f in (289, 805)] + \
             [(256 * n, 256, 2304, 9, 5, 1, f"{n} tiles 256x256, flags 5") for n in (8, 64, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "flags":
    # one tile per CU on 8 CUs: what each piece of the epilogue adds (0 = plain output, 1 = + bias, 33 = + second leaky-relu output, ...)
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    LAYERS = [(320 * nt, 256, 768, 3, f, 7, f"{nt} tiles, flags {f}") for f in (0, 1, 256 + 32, 289, 33, 5, 4 + 256 + 32, 805)]
for M, N, K, taps, flags, cfg, what in LAYERS:
    ms = T.time_cfg(M, N, K, 1, taps, flags, cfg, 4)
    T.time_cfg(M, N, K, 1, taps, flags, cfg, 1)                 # (the stamps are of the last launch)
    torch.cuda.synchronize()
    st = np.zeros(8192 * 6, dtype=np.uint64)
    assert lib.dmx_gemm_stamps_read(st.ctypes.data_as(C.c_void_p)) == 0
    bm, bn = TILE[cfg]
    nwg = min(8192, -(-M // bm) * -(-N // bn))
    st = st.reshape(8192, 6)[:nwg]
    t = st[:, :5].astype(np.float64) / 100.0
    ack = t[:, 4] - t[:, 3]
    t = np.concatenate([t[:, :3], t[:, 4:5]], axis=1)        # (entry, prologue, K loop, stores acknowledged)
    t -= t[:, 0].min()
    hw = st[:, 5]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7) + 128 * ((hw >> 12) & 1)   # xcc, cu_id, se_id, sh_id
    pro, kl, ep = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    flop_wg = 2.0 * bm * bn * K
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]; o = idx[np.argsort(t[idx, 0])]
        gaps += list(t[o[1:], 0] - t[o[:-1], 3])
    first = t[:, 0] < 2.0                                        # first round of workgroups
    print(f"{what:24s} M={M} N={N} K={K} cfg {cfg}: launch {ms*1e3:6.1f} us ({2.0*M*N*K/ms/1e9:6.0f} TF/s) wgs {nwg} on {len(np.unique(cu))} CUs | "
          f"prologue {pro.mean():5.2f}  K loop {kl.mean():6.2f} ({flop_wg/kl.mean()/1e6:5.2f} TF/s per CU)  epilogue {ep.mean():5.2f} "
          f"(of which waiting for the store acks {ack.mean():4.2f}; first round {ep[first].mean():5.2f}, later {ep[~first].mean() if (~first).any() else 0:5.2f})  life {(t[:,3]-t[:,0]).mean():6.2f}  "
          f"span {t[:,3].max():6.1f} | gap between wgs on a CU {np.mean(gaps) if gaps else 0:5.2f} | "
          f"epilogue start spread (first round) {t[first, 2].std():5.2f} us", flush=True)
