import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")): sys.path.insert(0, p)
from neutts import _hip
lib = _hip.load_library()
def probe(M, N, K, cfg, abl=48, copies=1, iters=20):
    us = C.c_double(); rc = lib.ntts_k_gemm_probe(M, N, K, cfg, abl, copies, iters, C.byref(us)); return us.value if rc == 0 else float("nan")
for name, (M, N, K) in {"pf_gate_up": (32000, 9728, 896), "pf_down": (32000, 896, 4864), "pf_qkv": (32000, 1152, 896), "codec_fc1": (65536, 4096, 1024), "codec_fc2": (65536, 1024, 4096)}.items():
    fl = 2.0 * M * N * K
    row = []
    for cfg, nm in ((42, "256x256 16w NS2 (today)"), (45, "256x256 8w NS2"), (48, "8w 4xK32"), (60, "8w 4xK32 STAG"), (61, "8w 3xK32 STAG"), (63, "16w 4xK32 STAG"), (64, "256x128 8w 4xK32 STAG")):
        t = min(probe(M, N, K, cfg) for _ in range(2))
        row.append(f"{nm}: {t:7.1f} us {fl / t / 1e6:5.0f} TF/s")
    print(f"== {name} M={M} N={N} K={K}: " + " | ".join(row), flush=True)
