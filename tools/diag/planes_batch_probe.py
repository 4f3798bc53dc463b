#!/usr/bin/env python3
"""A pool of device-resident 4K I420 surfaces -> JPEG files through jpegenc_encoder_encode_planes_batch_device (shared launches,
plane addresses through a device table), against one jpegenc_encoder_encode_planes_device call per frame and against the same
frames as interleaved RGB through jpegenc_encoder_encode_batch_device; the sink drops the bytes (the download is still made)."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h, n = 3840, 2160, 16
rgb = synth.test_img_rgb(w, h)
rng = np.random.default_rng(2)
frames, keep, rgbs = [], [], []
for f in range(n):
    px = np.clip(rgb.astype(np.int16) + rng.integers(-6, 7, rgb.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    # the I420 surface a camera / decoder would hand over for this RGB frame (image_buffer.rs:9-31 + 2x2 averaging), so both
    # ways code the same picture and their files have the same size
    r_, g_, b_ = (px[:, :, i].astype(np.int64) for i in range(3))
    yy = (19595 * r_ + 38470 * g_ + 7471 * b_ + 32767) >> 16
    cbf = (-11059 * r_ - 21709 * g_ + 32768 * b_ + (128 << 16) + 32767) >> 16
    crf = (32768 * r_ - 27439 * g_ - 5329 * b_ + (128 << 16) + 32767) >> 16
    avg = lambda a: ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    y = torch.from_numpy(np.ascontiguousarray(yy.astype(np.uint8))).cuda()
    cb = torch.from_numpy(np.ascontiguousarray(avg(cbf))).cuda(); cr = torch.from_numpy(np.ascontiguousarray(avg(crf))).cuda()
    keep += [y, cb, cr]
    frames.append([(y.data_ptr(), w, 1, 0), (cb.data_ptr(), w // 2, 1, 0), (cr.data_ptr(), w // 2, 1, 0)])
    rgbs.append(px)
d_rgb = torch.from_numpy(np.stack(rgbs)).cuda()
nbytes = [0]
def sink(user, ptr, k):
    nbytes[0] += k
    return 0
cb_ = b.WRITE_FN(sink)
users = (C.c_void_p * n)(*range(n))
arr = (b.Plane * (4 * n))()
for f, planes in enumerate(frames):
    for i, (ptr, pitch, stride, inv) in enumerate(planes):
        arr[4 * f + i] = b.Plane(ptr, pitch, stride, inv)
lib = b.lib()
fn = lib.jpegenc_encoder_encode_planes_batch_device
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(b.Plane), C.c_int, C.c_int, b.WRITE_FN, C.POINTER(C.c_void_p)]
one = lib.jpegenc_encoder_encode_planes_device
fb = lib.jpegenc_encoder_encode_batch_device
fb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, b.WRITE_FN, C.POINTER(C.c_void_p)]
e = b.Encoder(85); e.set_sampling_factor(b.F_2_2)
def timed(fn_, reps=5):
    fn_(); ts = []
    for _ in range(reps):
        nbytes[0] = 0; t = time.perf_counter(); fn_(); ts.append(time.perf_counter() - t)
    return min(ts)
t_batch = timed(lambda: b.check(fn(e._h, b.J_YCBCR, w, h, arr, n, 1, cb_, users)))
mb = nbytes[0] / n / 1e6
def each():
    for f in range(n):
        sub = (b.Plane * 4)(*[arr[4 * f + i] for i in range(4)])
        b.check(one(e._h, b.J_YCBCR, w, h, sub, 1, cb_, None))
t_each = timed(each)
e2 = b.Encoder(85); e2.set_sampling_factor(b.F_2_2)
t_rgb = timed(lambda: b.check(fb(e2._h, d_rgb.data_ptr(), w * h * 3, n, w, h, b.RGB, cb_, users)))
mb_rgb = nbytes[0] / n / 1e6
print(f"{n} 4K frames q85 4:2:0 in HBM -> files: I420 batch ({mb:.2f} MB each) {t_batch * 1e6 / n:6.1f} us/frame, I420 one call per frame {t_each * 1e6 / n:6.1f}, interleaved RGB batch ({mb_rgb:.2f} MB each) {t_rgb * 1e6 / n:6.1f}")
