#!/usr/bin/env python3
"""What hipHostRegister + one asynchronous upload + hipHostUnregister cost PER CALL on a caller's pageable buffer (us), for numpy arrays
(>= 4 MB: numpy asks for transparent huge pages), plain malloc'ed memory (4 KB pages under THP=madvise) and the same buffer again and again
against a fresh one every time - the alternative to staging single images through the library's own page-locked buffer."""
import ctypes as C, json, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
torch.zeros(1).cuda()
dst = torch.empty(140_000_000, dtype=torch.uint8, device="cuda")
stream = C.c_void_p()
hip.hipStreamCreateWithFlags(C.byref(stream), 1)
def cycle(ptr, n):
    t0 = time.perf_counter(); r = hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(n), 0)
    t1 = time.perf_counter(); hip.hipMemcpyAsync(C.c_void_p(dst.data_ptr()), C.c_void_p(ptr), C.c_size_t(n), 1, stream); hip.hipStreamSynchronize(stream)
    t2 = time.perf_counter(); u = hip.hipHostUnregister(C.c_void_p(ptr))
    t3 = time.perf_counter()
    return r, u, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
for n in (2_764_800, 6_220_800, 24_883_200, 132_710_400):
    for kind in ("numpy", "malloc"):
        for fresh in (False, True):
            rows = []
            keep = None
            for i in range(8):
                if fresh or keep is None:
                    if kind == "numpy":
                        arr = np.empty(n, dtype=np.uint8); arr[:] = i; ptr = arr.ctypes.data; keep = arr
                    else:
                        if keep is not None and fresh: libc.free(C.c_void_p(keep))
                        ptr = libc.malloc(n + 4096); C.memset(C.c_void_p(ptr), i, n); keep = ptr
                rows.append(cycle(ptr, n))
            if kind == "malloc": libc.free(C.c_void_p(keep))
            med = lambda k: sorted(r[k] for r in rows[2:])[len(rows[2:]) // 2]
            print(json.dumps({"bytes": n, "memory": kind, "buffer": "fresh every call" if fresh else "the same every call", "status": [rows[-1][0], rows[-1][1]],
                              "register_us": round(med(2)), "upload_us": round(med(3)), "unregister_us": round(med(4)), "link_only_us": round(n / 56e3)}), flush=True)
