#!/usr/bin/env python
"""Child process of tests/test_host_sanitized.py: sweeps the HOST side of libaesr_hip (planners, kernel selection, workspace arithmetic, item
tables, host-side job tables) in the AddressSanitizer + UBSan build of the library (csrc/Makefile target ``asan``), without a GPU.

    LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0 python tests/host_sanitized_sweep.py <libaesr_hip_asan.so> [seed] [count]

No torch, no device memory: queries are plain host functions; launch entry points get FAKE device pointers -- the host never dereferences a device
pointer, and on a box without a GPU a launch ends in a HIP error after all the host code in front of it (argument validation, plan_* with their
caches, ring / resident-filter item decomposition and its multiply-high magic numbers, slab and workspace sizes) has run.  A sanitizer report aborts
the process (non-zero exit); the parent asserts on the exit code and on the summary line."""
import ctypes
import random
import sys
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p

# fake device pointers must never reach a real device: this sweep only runs where the HIP runtime sees NO GPU (the parent hides them)
try:
    _hip = ctypes.CDLL("libamdhip64.so")
    _n = c_int(0)
    if _hip.hipGetDeviceCount(ctypes.byref(_n)) == 0 and _n.value > 0:
        print("REFUSED: %d GPU(s) visible -- the sanitized sweep drives launch entry points with fake pointers and runs on CPU-only boxes" % _n.value)
        sys.exit(3)
except OSError:
    pass
lib = ctypes.CDLL(sys.argv[1])
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
P, I, F, Z = c_void_p, c_int, c_float, c_size_t


def sig(name, res, args):
    f = getattr(lib, name)
    f.restype, f.argtypes = res, args
    return f


err = sig("aesr_last_error_string", c_char_p, [])
q = {n: sig(n, r, a) for n, r, a in [
    ("aesr_conv2d_wino_supported", I, [I] * 5), ("aesr_conv2d_wino_kernel", I, [I] * 8), ("aesr_conv2d_wino_workspace_floats", Z, [I] * 6),
    ("aesr_conv2d_wino_packed_floats", Z, [I] * 3), ("aesr_conv2d_packed_floats", Z, [I] * 4), ("aesr_conv2d_workspace_floats", Z, [I] * 7),
    ("aesr_conv2d_dgrad_workspace_floats", Z, [I] * 7), ("aesr_conv2d_wgrad_workspace_floats", Z, [I] * 7),
    ("aesr_conv2d_wino_fwd_bn_supported", I, [I] * 5), ("aesr_conv2d_wgrad_up2_supported", I, [I] * 2), ("aesr_bn_fused_supported", I, [I] * 2),
    ("aesr_bn_fused1_supported", I, [I] * 7), ("aesr_bn_fused1_workspace_floats", Z, [I] * 2), ("aesr_p2p_region_bytes", Z, [I]),
    ("aesr_ssim_workspace_doubles", Z, [I] * 3), ("aesr_vif_workspace_bytes", Z, [I] * 3), ("aesr_stemconv_folded_floats", Z, [I]),
    ("aesr_stemconv_workspace_floats", Z, [I]), ("aesr_conv2d_cout1_workspace_floats", Z, [I]), ("aesr_small_wgrad_workspace_floats", Z, [I])]}
run = {n: sig(n, r, a) for n, r, a in [
    ("aesr_conv2d_wino_fwd_ws", I, [P] * 5 + [Z] + [I] * 6 + [F, P]), ("aesr_conv2d_wino_dgrad_ws", I, [P] * 5 + [Z] + [I] * 6 + [F, P]),
    ("aesr_conv2d_wino_fwd_up2", I, [P] * 4 + [I] * 6 + [F, P]), ("aesr_conv2d_wino_dgrad_sum2", I, [P] * 3 + [I] * 5 + [P]),
    ("aesr_conv2d_wino_fwd_bn", I, [P] * 6 + [I] * 6 + [F, I, P]),
    ("aesr_conv2d_fwd_ws", I, [P] * 5 + [I] * 8 + [F, P]), ("aesr_conv2d_dgrad_ws", I, [P] * 5 + [I] * 8 + [F, P]),
    ("aesr_conv2d_wgrad_partial", I, [P] * 3 + [I] * 8 + [P]), ("aesr_conv2d_wgrad", I, [P] * 5 + [I] * 7 + [P]),
    ("aesr_bn_apply", I, [P] * 4 + [I] * 6 + [ctypes.POINTER(c_int), P]),
    ("aesr_lerp_multi", I, [P, P, I, Z, ctypes.POINTER(c_float), I, I, F, P])]}

FAKE = [0x7f0000000000 + (k << 32) for k in range(8)]          # never dereferenced on the host
rng = random.Random(seed)
BASELINE = [(n, h, ci, co) for n in (1, 2, 3, 4, 6, 12, 24, 36, 48) for (h, ci, co) in
            [(162, 32, 32), (81, 32, 64), (81, 64, 64), (40, 64, 128), (40, 128, 128), (40, 128, 64), (40, 64, 64), (80, 64, 32), (80, 32, 32), (160, 32, 32),
             (222, 32, 32), (111, 64, 64), (55, 128, 128), (258, 32, 32), (129, 64, 64), (64, 128, 128), (20, 256, 256), (20, 128, 256),
             (160, 64, 64), (80, 128, 128), (40, 256, 256), (20, 512, 512), (10, 512, 512), (220, 64, 64), (110, 128, 128), (55, 256, 256), (27, 512, 512),
             (13, 512, 512), (256, 64, 64), (128, 128, 128), (64, 256, 256), (32, 512, 512), (16, 512, 512), (28, 16, 16), (7, 32, 16), (30, 8, 8)]]


# tensors beyond the kernels' 32-bit offsets: the entry points must REFUSE them, and the arithmetic that decides so must not itself overflow
HUGE = [(64, 1024, 1024, 64, 64), (512, 512, 512, 32, 32), (3, 16384, 16384, 32, 32), (2048, 162, 162, 32, 32), (1, 30000, 30000, 16, 32), (100000, 8, 8, 512, 512),
        (16, 2048, 2048, 128, 128), (2147483647, 1, 1, 32, 32), (1, 46341, 46341, 32, 32)]


def shapes():
    for s in BASELINE:
        yield s[0], s[1], s[1], s[2], s[3]
    for s in HUGE:
        yield s
    for _ in range(count):
        n = rng.choice([1, 1, 2, 3, 5, 7, 12, 16, 33])
        h, w = rng.randint(1, 300), rng.randint(1, 300)
        if rng.random() < 0.3:
            w = h
        ci = rng.choice([1, 3, 4, 8, 16, 32, 48, 64, 96, 128, 256, 512])
        co = rng.choice([1, 4, 8, 16, 32, 64, 96, 128, 256, 512])
        if n * h * w * max(ci, co) > (1 << 29):
            continue
        yield n, h, w, ci, co


ncalls = nrun = 0
status = {}
for n, h, w, ci, co in shapes():
    for tr in (0, 1):
        ok = q["aesr_conv2d_wino_supported"](ci, co, 3, 1, tr)
        q["aesr_conv2d_wino_packed_floats"](co, ci, tr)
        for ks in (1, 3):
            q["aesr_conv2d_packed_floats"](co, ci, ks, tr)
        ncalls += 4
        if ok:
            kern = q["aesr_conv2d_wino_kernel"](n, h, w, ci, co, 3, 1, tr)
            nws = q["aesr_conv2d_wino_workspace_floats"](n, h, w, ci, co, tr)
            big = n * (h + 2) * (w + 2) * max(ci, co) >= 0x1C000000 or n > (1 << 24) or max(h, w) > (1 << 15)
            assert (kern == 0 and nws == 0) if big else kern in (1, 2, 3), (kern, nws, n, h, w, ci, co)       # refused, not planned
            assert nws == 0 or nws % (n * h * w * (ci if tr else co)) == 0, (nws, n, h, w, ci, co, tr)
            f = run["aesr_conv2d_wino_dgrad_ws" if tr else "aesr_conv2d_wino_fwd_ws"]
            for ws in ((P(FAKE[4]), nws), (None, 0)):
                rc = f(P(FAKE[0]), P(FAKE[1]), P(FAKE[2]) if rng.random() < 0.7 else None, P(FAKE[3]), ws[0], ws[1], n, h, w, ci, co,
                       rng.choice([0, 1, 2]), 0.01, None)
                status[rc] = status.get(rc, 0) + 1
                nrun += 1
            ncalls += 2
    ks = rng.choice([1, 3])
    pad = ks // 2
    for name in ("aesr_conv2d_workspace_floats", "aesr_conv2d_dgrad_workspace_floats", "aesr_conv2d_wgrad_workspace_floats"):
        q[name](n, h, w, ci, co, ks, pad)
        ncalls += 1
    if ci % 4 == 0 and co % 4 == 0:
        nws = q["aesr_conv2d_workspace_floats"](n, h, w, ci, co, ks, pad)
        rc = run["aesr_conv2d_fwd_ws"](P(FAKE[0]), P(FAKE[1]), P(FAKE[2]), P(FAKE[3]), P(FAKE[4]) if nws else None, n, h, w, ci, co, ks, pad, 1, 0.01, None)
        status[rc] = status.get(rc, 0) + 1
        nws = q["aesr_conv2d_dgrad_workspace_floats"](n, h, w, ci, co, ks, pad)
        rc = run["aesr_conv2d_dgrad_ws"](P(FAKE[0]), P(FAKE[1]), P(FAKE[2]), P(FAKE[3]), P(FAKE[4]) if nws else None, n, h, w, ci, co, ks, pad, 1, 0.01, None)
        status[rc] = status.get(rc, 0) + 1
        rc = run["aesr_conv2d_wgrad_partial"](P(FAKE[0]), P(FAKE[1]), P(FAKE[4]), n, h, w, ci, co, ks, pad, 0, None)
        status[rc] = status.get(rc, 0) + 1
        nrun += 3
    if q["aesr_conv2d_wgrad_up2_supported"](ci, co) and h % 2 == 0 and w % 2 == 0:
        for name, args in (("aesr_conv2d_wino_fwd_up2", [P(FAKE[0]), P(FAKE[1]), P(FAKE[2]), P(FAKE[3]), n, h, w, ci, co, 1, 0.01, None]),
                           ("aesr_conv2d_wino_dgrad_sum2", [P(FAKE[0]), P(FAKE[1]), P(FAKE[3]), n, h, w, ci, co, None])):
            rc = run[name](*args)
            status[rc] = status.get(rc, 0) + 1
            nrun += 1
    if q["aesr_conv2d_wino_fwd_bn_supported"](n, h, w, ci, co):
        rc = run["aesr_conv2d_wino_fwd_bn"](P(FAKE[0]), P(FAKE[1]), P(FAKE[2]), P(FAKE[5]), P(FAKE[6]), P(FAKE[3]), n, h, w, ci, co, 1, 0.01, rng.choice([0, 1]), None)
        status[rc] = status.get(rc, 0) + 1
        nrun += 1
    for g in (1, 2, 3):
        q["aesr_bn_fused_supported"](co, g)
        for mode in (0, 1, 2):
            for bwd in (0, 1):
                q["aesr_bn_fused1_supported"](n, h, w, co, mode, g, bwd)
        q["aesr_bn_fused1_workspace_floats"](co, g)
        ncalls += 8
    if co % 4 == 0:
        ns = (c_int * 3)(0, max(1, n // 2), n)
        rc = run["aesr_bn_apply"](P(FAKE[0]), P(FAKE[1]), P(FAKE[2]), P(FAKE[3]), n, h, w, co, rng.choice([0, 1, 2]), 2 if n > 1 else 1, ns, None)
        status[rc] = status.get(rc, 0) + 1
        nrun += 1
    q["aesr_ssim_workspace_doubles"](n, h, w)
    q["aesr_vif_workspace_bytes"](n, h, w)
    for c in (ci, co):
        q["aesr_stemconv_folded_floats"](c), q["aesr_stemconv_workspace_floats"](c), q["aesr_conv2d_cout1_workspace_floats"](c), q["aesr_small_wgrad_workspace_floats"](c)
    ncalls += 10
for world in range(-1, 11):
    q["aesr_p2p_region_bytes"](world)
al = (c_float * 5)(0.2, 0.4, 0.5, 0.6, 0.8)
for z in (2, 3, 30):
    rc = run["aesr_lerp_multi"](P(FAKE[0]), P(FAKE[1]), z, 64 * 56 * 56, al, 5, 1, 0.01, None)
    status[rc] = status.get(rc, 0) + 1
    nrun += 1
assert 0 not in status, "a launch reported success on a box without a GPU: %r" % status        # (on a GPU box this sweep is not meant to run)
print("SANITIZED SWEEP OK: %d host queries, %d entry points driven up to their launch (status histogram %r); last library message: %s"
      % (ncalls, nrun, status, (err() or b"").decode()[:100]))
